#!/usr/bin/env python
"""GEMM kernel A/B on the GPU box (random data, the engine's epilogues):
  python scripts/gemm_bench.py                 gemm2 tuning switches side by side on the r640 1-hour shapes
  python scripts/gemm_bench.py variants 1 2    gemm.hip (128x128) vs gemm2.hip (256x256 LDS-DMA)
Opts are (flags, group_m): flags bit 0 = 32x32x16 MFMAs, bit 1 = s_setprio for waves 4-7; group_m = tile order."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reverb_amd import _lib

lib = _lib.load_test()
SHAPES = [  # (M, N, K, act, out_f32, with_res, label)  -- the r640 1-hour workload (first slice: 144 chunks)
    (73728, 4096, 1024, 1, 0, 0, "ffn1"), (73728, 1024, 4096, 0, 1, 1, "ffn2"), (73728, 3072, 1024, 0, 0, 0, "qkv"),
    (73728, 1024, 1024, 0, 1, 1, "out/pw2"), (73728, 2048, 1024, 0, 0, 0, "pw1"), (8192, 10001, 1024, 0, 1, 0, "ctc slab"),
    (73728, 1024, 19456, 0, 1, 0, "embed"), (16384, 4096, 1024, 1, 0, 0, "ffn1 32ch"), (16384, 1024, 1024, 0, 1, 1, "out 32ch"),
    (142560, 1024, 1024, 0, 0, 0, "dec q"), (142560, 4096, 1024, 2, 0, 0, "dec ff1"),
]


def run(dtype, M, N, K, variant, iters, act, of32, res):
    ms, md = C.c_double(0), C.c_double(0)
    rc = lib.rvb_test_gemm_bench(dtype, M, N, K, variant, iters, act, of32, res, C.byref(ms), C.byref(md))
    if rc != 0:
        return None, lib.rvb_last_error().decode()
    return ms.value, md.value


if len(sys.argv) > 1 and sys.argv[1] == "variants":
    variants = [int(v) for v in sys.argv[2:]] or [1, 2]
    for dtype, name in [x for x in ((1, "bf16"), (0, "f32")) if x[1] in os.environ.get("DTYPES", "bf16,f32")]:
        for (M, N, K, act, of32, res, label) in SHAPES:
            if dtype == 0 and M > 30000:
                M = 22528
            line = f"{name} {label:10s} M={M} N={N} K={K}:"
            for v in variants:
                ms, md = run(dtype, M, N, K, v, 5 if dtype else 2, act, of32, res)
                line += f"  v{v}: ERR {md}" if ms is None else f"  v{v}: {ms:8.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s (maxdiff {md:.2e})"
            print(line, flush=True)
    sys.exit(0)

OPTS = [(0, 0), (1, 0), (0, 8), (1, 8), (3, 8), (1, 4), (1, 16), (2, 0)]
if len(sys.argv) > 1:
    OPTS = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
tot = {o: 0.0 for o in OPTS}
print("opts (flags,group_m): " + "  ".join(str(o) for o in OPTS))
for (M, N, K, act, of32, res, label) in SHAPES:
    line = f"bf16 {label:10s} M={M:6d} N={N:5d} K={K:5d}:"
    for o in OPTS:
        lib.rvb_test_set_gemm2_opts(o[0], o[1])
        ms, md = run(1, M, N, K, 2, 8, act, of32, res)
        if ms is None:
            line += f"  ERR {md}"
            continue
        tot[o] += ms
        line += f"  {2.0 * M * N * K / ms / 1e9:7.1f}" + ("" if md < 0.1 else f"(!diff {md:.1e})")
    print(line, flush=True)
print("sum ms:" + "  ".join(f"{o}: {tot[o]:.3f}" for o in OPTS))
lib.rvb_test_set_gemm2_opts(-1, -1)
