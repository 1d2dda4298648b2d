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


def mp3_seeds():
    """Layer III streams of tests/mp3_writer.py (MPEG-1 / 2 / 2.5, joint stereo, block switching, CRC, reservoir) + the real file;
    the synthesis window the encoder's analysis needs is read from the lab library (built by `python -m reverb_amd.build`)."""
    import ctypes as C
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(here, "tests"))
    import mp3_writer as Wr
    G = Wr.GranuleCfg
    lib = C.CDLL(os.path.join(here, "reverb_amd", "librvb_test.so"))
    w = np.zeros(512, np.float32)
    lib.rvb_test_mp3_window(w.ctypes.data_as(C.POINTER(C.c_float)))
    w = w.astype(np.float64)
    rng = np.random.default_rng(5)
    bt = lambda g: [0, 0, 1, 2, 2, 3, 0, 1, 2, 3][g % 10]

    def sig(n, nch):
        t = np.arange(n)
        return np.stack([0.1 * np.sin(0.05 * (c + 1) * t) + 0.02 * rng.standard_normal(n) for c in range(nch)])

    out = [Wr.Encoder(44100, 2, 128, w, mode=1, mode_ext=2, plan=lambda g: G(bt(g))).encode(sig(1152 * 4, 2)),
           Wr.Encoder(48000, 1, 96, w, scfsi=True, crc=True, seed=2).encode(sig(1152 * 4, 1)),
           Wr.Encoder(22050, 2, 64, w, plan=lambda g: G(bt(g), bt(g) != 0 and g % 2 == 0), seed=3).encode(sig(576 * 6, 2)),
           Wr.Encoder(8000, 1, 32, w, plan=lambda g: G(bt(g)), seed=4).encode(sig(576 * 6, 1)),
           Wr.Encoder(44100, 1, 64, w, bit_share=lambda f: 0.25 if f % 3 else 3.0, seed=7).encode(sig(1152 * 6, 1))]
    real = open(os.path.join(here, "tests", "golden", "mathjax_invalid_keypress.mp3"), "rb").read()
    out.append(real)                 # 43 frames: enough for decode() to split it over up to 5 threads
    return [bytes(x) for x in out]


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
    if os.environ.get("FUZZ_MP3", "1") != "0":
        base = base + mp3_seeds()
    with open(path, "wb") as f:
        for it in range(n):
            d = mutate(base[it % len(base)], rng)
            f.write(struct.pack("<I", len(d)) + d)


if __name__ == "__main__":
    main()
