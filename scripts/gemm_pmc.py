#!/usr/bin/env python
"""Two GEMM shapes of the r640 workload under `rocprofv3 --pmc ...` (see profiles/)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reverb_amd import _lib
lib = _lib.load_test()
for (M, N, K, act, of32, res) in [(90112, 3072, 1024, 0, 0, 0), (90112, 1024, 1024, 0, 1, 1), (90112, 1024, 19456, 0, 1, 0)]:
    ms = C.c_double(0)
    rc = lib.rvb_test_gemm_bench(1, M, N, K, 2, 2, act, of32, res, C.byref(ms), None)
    print(M, N, K, rc, ms.value, 2.0 * M * N * K / max(ms.value, 1e-9) / 1e9, flush=True)
