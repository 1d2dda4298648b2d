#!/usr/bin/env python
"""GEMM kernel A/B on the GPU box: python scripts/gemm_bench.py  (prints TFLOP/s per shape/variant)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reverb_amd import _lib

lib = _lib.load()
SHAPES = [  # (M, N, K, act, out_f32, with_res, label)  -- the r640 1-hour workload
    (90112, 4096, 1024, 1, 0, 0, "ffn1"), (90112, 1024, 4096, 0, 1, 1, "ffn2"), (90112, 3072, 1024, 0, 0, 0, "qkv"),
    (90112, 1024, 1024, 0, 1, 1, "out/pw2"), (90112, 2048, 1024, 0, 0, 0, "pw1"), (8192, 10001, 1024, 0, 1, 0, "ctc slab"),
    (90112, 1024, 19456, 0, 1, 0, "embed"), (22528, 4096, 1024, 1, 0, 0, "ffn1 0.25h"),
]
variants = [int(v) for v in sys.argv[1:]] or [1, 2]
for dtype, name in [x for x in ((1, "bf16"), (0, "f32")) if x[1] in os.environ.get("DTYPES", "bf16,f32")]:
    for (M, N, K, act, of32, res, label) in SHAPES:
        if dtype == 0 and M > 30000:
            M = 22528
        line = f"{name} {label:10s} M={M} N={N} K={K}:"
        for v in variants:
            ms, md = C.c_double(0), C.c_double(0)
            rc = lib.rvb_test_gemm_bench(dtype, M, N, K, v, 5 if dtype else 2, act, of32, res, C.byref(ms), C.byref(md))
            if rc != 0:
                line += f"  v{v}: ERR {lib.rvb_last_error().decode()}"
                continue
            line += f"  v{v}: {ms.value:8.3f} ms {2.0 * M * N * K / ms.value / 1e9:7.1f} TF/s (maxdiff {md.value:.2e})"
        print(line, flush=True)
