"""Mutation corpus for scripts/fuzz/audio_harness.cpp: records of [u32 length][bytes]."""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import flac_writer as FW                                              # noqa: E402
from tests.test_audio_decode import RFC9639_D1, aiff_bytes, signal, wav_bytes    # noqa: E402


def seeds():
    x, y, z = signal(2, 3000, 16, 1), signal(1, 2000, 24, 2), signal(2, 500, 8, 3)
    return [FW.encode(x, 16, 16000, block=1024, stereo="mid_side", predictor=("lpc", 8, 12)),
            FW.encode(x, 16, 16000, block=256, stereo="left_side"),
            FW.encode(y, 24, 48000, block=500, predictor=("fixed", 3), escape_first=True),
            FW.encode(z, 8, 8000, block=64, stereo="side_right", md5=False, total_known=False),
            FW.encode(x, 16, 16000, variable_blocks=[1000, 16, 984, 1000], id3=True), RFC9639_D1,
            wav_bytes(1, 2, 16000, 16, x.T.astype("<i2").tobytes()), wav_bytes(3, 1, 16000, 32, np.zeros(100, "<f4").tobytes()),
            wav_bytes(6, 1, 8000, 8, bytes(range(256))), wav_bytes(1, 2, 8000, 24, bytes(600), extensible=True),
            aiff_bytes(2, 16000, 16, x.T.astype(">i2").tobytes()), aiff_bytes(1, 8000, 8, bytes(200), compression=b"NONE"),
            aiff_bytes(1, 22050, 32, np.zeros(50, ">f4").tobytes(), compression=b"fl32"),
            # odd-sized trailing chunks without their pad byte (the round-3 walker ran past the buffer on these two)
            _riff(b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"LIST" + struct.pack("<I", 3) + b"abc"),
            b"FORM" + struct.pack(">I", 17) + b"AIFF" + b"ANNO" + struct.pack(">I", 5) + b"hello"]


def _riff(body):
    return b"RIFF" + struct.pack("<I", len(body)) + body


def mutate(d, rng):
    d = bytearray(d)
    for _ in range(int(rng.integers(1, 5))):
        if len(d) < 2:
            break
        mode = int(rng.integers(0, 4))
        if mode == 0:
            d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        elif mode == 2:
            cut = int(rng.integers(0, len(d)))
            d = d[:cut] if rng.integers(0, 2) else d[:cut] + d[cut + int(rng.integers(1, 50)):]
        else:
            p = int(rng.integers(0, max(1, len(d) - 4)))
            d[p:p + 4] = struct.pack("<I", int(rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, len(d)])))
    return bytes(d)


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    rng, base = np.random.default_rng(1), seeds()
    with open(path, "wb") as f:
        for it in range(n):
            d = mutate(base[it % len(base)], rng)
            f.write(struct.pack("<I", len(d)) + d)


if __name__ == "__main__":
    main()
