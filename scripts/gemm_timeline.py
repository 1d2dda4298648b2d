"""Where a gemm2 tile's time goes: per-workgroup timestamps (rvb_test_gemm_timeline) -> phase durations, tiles per CU,
how many workgroups sit in their epilogue at the same time."""
import ctypes as C, sys
import numpy as np
from reverb_amd import _lib
lib = _lib.load_test()
M = 73728
for name, N, K, act, of32, res in (("ffn1", 4096, 1024, 1, 0, 0), ("out/pw2", 1024, 1024, 0, 1, 1), ("ffn2", 1024, 4096, 0, 1, 1),
                                   ("qkv", 3072, 1024, 0, 0, 0), ("plain N=K=1024 bf16 out", 1024, 1024, 0, 0, 0), ("N=K=1024 fp32 out, no residual", 1024, 1024, 0, 1, 0)):
    cap = (M // 256 + 1) * (N // 256 + 1)
    out = np.zeros(cap * 6, np.int64)
    n = C.c_int32(0)
    _lib.check(lib.rvb_test_gemm_timeline(M, N, K, act, of32, res, out.ctypes.data_as(C.POINTER(C.c_longlong)), cap, C.byref(n)))
    t = out[:n.value * 6].reshape(-1, 6)
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0
    pro, main, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    total = us[:, 3].max()
    hw = t[:, 4]; xcc = t[:, 5] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0x3) << 7) | (xcc << 9)     # cu_id, sh_id, se_id, xcc
    ncu = len(np.unique(cu))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    flops = 2.0 * M * N * K
    print(f"== {name}: M={M} N={N} K={K}  {n.value} tiles on {ncu} CUs ({per_cu.min()}..{per_cu.max()} per CU), launch {total:.1f} us = {flops / total / 1e6:.0f} TFLOP/s")
    for lbl, v in (("prologue", pro), ("main loop", main), ("epilogue", epi)):
        print(f"   {lbl:9s} mean {v.mean():6.2f} us  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
    print(f"   per K step (main / {K // 64}): {main.mean() / (K // 64):.3f} us;  busy sum per CU {(us[:, 3] - us[:, 0]).sum() / ncu:.1f} us of {total:.1f}")
    # concurrency: how many workgroups are in each phase over time
    grid = np.linspace(0, total, 41)[1:-1]
    inepi = [(int(((us[:, 2] <= g) & (us[:, 3] > g)).sum()), int(((us[:, 1] <= g) & (us[:, 2] > g)).sum())) for g in grid]
    print("   (in epilogue, in main loop) over time:", " ".join(f"{a}/{b}" for a, b in inepi[::3]))
    # gaps between consecutive workgroups on one CU
    gaps = []
    for c in np.unique(cu):
        w = np.sort(us[cu == c][:, [0, 3]], axis=0)
        gaps += list(w[1:, 0] - w[:-1, 1])
    if gaps:
        print(f"   gap between a CU's consecutive workgroups: mean {np.mean(gaps):.2f} us  p90 {np.percentile(gaps, 90):.2f}")
