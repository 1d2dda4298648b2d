"""Minimal RIFF/WAVE reader for the front end (the reference uses torchaudio.load(normalize=False),
asr/wenet/cli/reverb.py:128, which is not available here).  16-bit PCM only; other encodings and
sample rates are the "next" row of SURVEY.md 8(f)."""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """Returns (int16 array of shape (channels, samples), sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE) or bits != 16:
        raise ValueError(f"{path}: only 16-bit PCM WAV is supported (format tag {tag}, {bits} bits)")
    n = len(pcm) // (2 * channels)
    arr = np.frombuffer(pcm[:n * 2 * channels], dtype="<i2").reshape(n, channels).T
    return np.ascontiguousarray(arr), rate
