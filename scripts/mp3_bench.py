#!/usr/bin/env python
"""Decode speed of csrc/mp3.cpp on this machine's host cores: a 44.1 kHz joint-stereo stream written by tests/mp3_writer.py (pure Python: the
encoding is the slow part of this script), decoded with 1 .. 16 threads through the test hook; the result is the same array whatever
the thread count (tests/test_mp3.py)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mp3_tables as MT        # noqa: E402
import mp3_writer as Wr        # noqa: E402
from reverb_amd import _lib    # noqa: E402

lib = _lib.load_test()
D = np.array(MT.D) / 65536.0
rng = np.random.default_rng(0)
frames = int(os.environ.get("MP3_BENCH_FRAMES", "1200"))
n = 1152 * frames
t = np.arange(n) / 44100
x = np.stack([0.2 * np.sin(2 * np.pi * 440 * t) + 0.02 * rng.standard_normal(n), 0.2 * np.sin(2 * np.pi * 550 * t) + 0.02 * rng.standard_normal(n)])
t0 = time.time()
data = Wr.Encoder(44100, 2, 128, D, mode=1, mode_ext=2).encode(x)
print(f"{n / 44100:.1f} s of 44.1 kHz joint stereo at 128 kbit/s: {len(data)} bytes in {frames} frames (encoded in {time.time() - t0:.0f} s by the test writer); host cpus {os.cpu_count()}")
info, st = (C.c_int64 * 9)(), (C.c_int64 * 12)()
out = np.zeros((2, n), np.float32)
ref = None
for nt in (1, 2, 4, 8, 16):
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        r = lib.rvb_test_mp3_decode(data, len(data), -1, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, info, st, nt)
        best = min(best, time.time() - t0)
    assert r == n
    if ref is None:
        ref = out.copy()
    same = bool(np.array_equal(ref, out))
    print(f"{nt:2d} threads: {best * 1e3:7.1f} ms = {n / 44100 / best:7.0f} x real time (both channels); identical to 1 thread: {same}")
t0 = time.time()
lib.rvb_test_mp3_decode(data, len(data), 0, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, info, st, 16)
dt = time.time() - t0
print(f"channel 0 only (what the ASR front end asks for), 16 threads: {dt * 1e3:.1f} ms = {n / 44100 / dt:.0f} x real time")
