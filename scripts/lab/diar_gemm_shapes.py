#!/usr/bin/env python
"""Lab: the segmentation network's GEMM shapes (LSTM input projections: M = windows x 293 frames, N = 1024, K = 64 / 256) on gemm2's
variants: flags 0 = the phase-interleaved loop, 16 = its persistent form with cross-tile prefetch (needs M % 256 == 0), 4 = the
register-pipelined loop; and the 128 x 128 kernel of gemm.hip (variant 1)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reverb_amd import _lib
lib = _lib.load_test()


def run(M, N, K, variant, iters=6):
    ms, md = C.c_double(0), C.c_double(0)
    rc = lib.rvb_test_gemm_bench(1, M, N, K, variant, iters, 0, 0, 0, C.byref(ms), C.byref(md))
    return (ms.value, md.value) if rc == 0 else (None, lib.rvb_last_error().decode())


for (M, N, K) in ((1052160, 1024, 256), (1052163, 1024, 256), (1052163, 1024, 64)):
    line = f"M={M} N={N} K={K}: bytes {(M * K + M * N) * 2 / 1e9:.2f} GB"
    for flags in (0, 16, 4):
        lib.rvb_test_set_gemm2_opts(flags, 0)
        ms, md = run(M, N, K, 2)
        line += f" | flags {flags}: " + (f"ERR {md}" if ms is None else f"{ms:.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF/s {(M * K + M * N) * 2 / ms / 1e9:5.2f} TB/s")
    lib.rvb_test_set_gemm2_opts(-1, -1)
    ms, md = run(M, N, K, 1)
    line += " | 128x128: " + (f"ERR {md}" if ms is None else f"{ms:.3f} ms")
    print(line, flush=True)
