"""Per-kernel parity tests: each HIP kernel (called through the C ABI's rvb_test_* entry points)
against a plain torch/numpy fp64 reference of the same reference op.  Tolerances are written at
each assert: f32 mode differs from the reference only by summation order; bf16 mode additionally
rounds operands/outputs to bfloat16."""
import math

import numpy as np
import pytest
import torch

from reverb_amd import _lib
from reverb_amd._lib import fptr, iptr, dptr
from util import bf16_round, rnd, f32, i32

pytestmark = pytest.mark.gpu
F32, BF16 = 0, 1


def _tol(dtype, f32_tol, bf16_tol):
    return f32_tol if dtype == F32 else bf16_tol


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 96), (70, 1001, 160), (513, 48, 32), (300, 200, 128), (513, 330, 192),
                                   (1000, 1001, 256)])
def test_gemm_plain(lib, dtype, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rnd(dtype, rng.standard_normal((M, K)))
    W = rnd(dtype, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    C = np.empty((M, N), np.float32)
    _lib.check(lib.rvb_test_gemm(dtype, fptr(A), fptr(W), fptr(bias), None, fptr(C), M, N, K, 1.0, 0, 1, 0, 0, 0, 0, 0))
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    # asymmetric random operands: a transposed fragment mapping cannot pass this
    np.testing.assert_allclose(C, ref, rtol=_tol(dtype, 2e-5, 2e-5), atol=_tol(dtype, 2e-5, 2e-5) * 4)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("act,alpha,use_res,out_f32", [(1, 1.0, False, 0), (2, 1.0, False, 0), (0, 0.5, True, 1), (0, 8.0, False, 1)])
def test_gemm_epilogue(lib, dtype, act, alpha, use_res, out_f32):
    M, N, K = 200, 136, 64
    rng = np.random.default_rng(7)
    A = rnd(dtype, rng.standard_normal((M, K)))
    W = rnd(dtype, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    res = f32(rng.standard_normal((M, N))) if use_res else None
    C = np.empty((M, N), np.float32)
    _lib.check(lib.rvb_test_gemm(dtype, fptr(A), fptr(W), fptr(bias), fptr(res), fptr(C), M, N, K, alpha, act, out_f32,
                                 0, 0, 0, 0, 0))
    v = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if act == 1:
        v = v / (1 + np.exp(-v))
    elif act == 2:
        v = np.maximum(v, 0)
    v = v * alpha + (res if use_res else 0)
    if dtype == BF16 and not out_f32:
        np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)      # one bf16 rounding of the output
    else:
        np.testing.assert_allclose(C, v, rtol=5e-5, atol=2e-4)


@pytest.mark.parametrize("flags,group_m", [(0, 0), (1, 0), (1, 8), (3, 4), (0, 8), (1, 3), (4, 8), (4, 0), (0, -2), (32, -2), (32, 0), (1024, -2), (2048, 0), (1025, 0), (8192, -2), (8192, 0)])
@pytest.mark.parametrize("M,N,K,act,alpha,use_res,out_f32", [
    (300, 200, 128, 0, 1.0, False, 1),        # one partial tile, aligned rows: LDS-transposed vector epilogue
    (513, 330, 192, 1, 1.0, False, 0),        # unaligned bf16 rows: element-wise epilogue, SiLU
    (1000, 1024, 256, 0, 0.5, True, 1),       # 4 x 4 tiles, fp32 residual
    (2100, 520, 64, 2, 1.0, False, 0),        # 9 x 3 tiles (ragged last column tile), one K step, ReLU, bf16 out
    (2600, 1280, 320, 0, 1.0, True, 1),       # 11 x 5 tiles: grouped orders with a ragged last group
    (700, 520, 128, 0, 0.5, True, 1),         # residual preloaded into the accumulators with a ragged last column tile and row tile
    (515, 264, 64, 0, 1.0, True, 0),          # residual + bf16 output, one K step
    (300, 262, 64, 0, 1.0, True, 1),          # N % 4 != 0: the residual stays in the (element-wise) epilogue
])
def test_gemm2_tuning_switches_keep_results(lib, flags, group_m, M, N, K, act, alpha, use_res, out_f32):
    """gemm2.hip's tuning switches (32x32x16 MFMAs with their own fragment / accumulator / epilogue mapping, grouped
    tile order, wave priority; flags 0 = the phase-interleaved loop of round 3, bit 2 = the register-pipelined loop of
    round 2, bit 5 = the residual added in the epilogue from a ring of prefetched vectors, group_m -2 = the per-shape
    default; bit 10 = full tiles through the generic epilogue instead of the one with inline-asm stores, bit 11 = residual
    tiles through it only for short K) against fp64 on asymmetric random operands, every epilogue form.  (The first form of
    the inline-asm stores left ONE wait state between a 16-byte store and the next VALU write of its data registers: 16-128
    wrong elements per million in the 32x32 kernel with a residual -- this test found them; gfx950 wants two.)"""
    dtype = BF16
    rng = np.random.default_rng(M + N + K)
    A = rnd(dtype, rng.standard_normal((M, K)))
    W = rnd(dtype, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    res = f32(rng.standard_normal((M, N))) if use_res else None
    C = np.full((M, N), np.nan, np.float32)
    lib.rvb_test_set_gemm2_opts(flags, group_m)
    try:
        _lib.check(lib.rvb_test_gemm(dtype, fptr(A), fptr(W), fptr(bias), fptr(res), fptr(C), M, N, K, alpha, act, out_f32,
                                     0, 0, 0, 0, 0))
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)
    v = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if act == 1:
        v = v / (1 + np.exp(-v))
    elif act == 2:
        v = np.maximum(v, 0)
    v = v * alpha + (res if use_res else 0)
    if out_f32:
        np.testing.assert_allclose(C, v, rtol=5e-5, atol=2e-4)
    else:
        np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)      # one bf16 rounding of the output


@pytest.mark.parametrize("flags", [0, 8192, 1024])
@pytest.mark.parametrize("M,N,K,rows,col0,cols", [
    (1024, 768, 128, 512, 256, 256),      # the engine's form in small: q | k | v thirds, whole tiles inside the k third, rows = frames per chunk
    (1536, 3072, 1024, 512, 1024, 1024),  # the r640 qkv GEMM (three chunks): twelve column tiles, four of them with the addend
    (700, 520, 64, 300, 128, 200),        # ragged tiles, a column range that cuts through tiles, rows that wrap inside a tile
    (130, 72, 64, 50, 8, 40),             # small: the 128x128 kernel
    (1024, 512, 128, 97, 256, 256),       # a row period that is no multiple of anything
])
def test_gemm_row_periodic_addend(lib, flags, M, N, K, rows, col0, cols):
    """GemmArgs::rowadd (round 6): the qkv GEMM of an encoder block writes K' = k + p -- p = the layer's positional keys, one bf16
    row per frame of a chunk -- in its epilogue, in fp32 before the one rounding to bf16.  Against fp64 on every path: full tiles
    (the epilogue with inline-asm stores and the prefetched addend ring), ragged tiles / a range cutting through a tile (the
    generic loop), the small-shape kernel; on the 16x16x32 and the 32x32x16 form of the loop and with the fast epilogue off."""
    rng = np.random.default_rng(M + N + rows)
    A = rnd(BF16, rng.standard_normal((M, K)))
    W = rnd(BF16, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    add = rnd(BF16, 3.0 * rng.standard_normal((rows, cols)))
    C = np.full((M, N), np.nan, np.float32)
    lib.rvb_test_set_gemm2_opts(flags, -2)
    try:
        _lib.check(lib.rvb_test_gemm_rowadd(fptr(A), fptr(W), fptr(bias), fptr(add), fptr(C), M, N, K, rows, col0, cols))
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)
    v = A.astype(np.float64) @ W.astype(np.float64).T + bias
    v[:, col0:col0 + cols] += add[np.arange(M) % rows]
    np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)            # one bf16 rounding of the output
    # ... and nothing leaked outside the column range: there the result equals the plain GEMM's exactly
    P = np.empty((M, N), np.float32)
    _lib.check(lib.rvb_test_gemm(BF16, fptr(A), fptr(W), fptr(bias), None, fptr(P), M, N, K, 1.0, 0, 0, 0, 0, 0, 0, 0))
    outside = np.ones(N, bool); outside[col0:col0 + cols] = False
    assert np.array_equal(C[:, outside], P[:, outside])


@pytest.mark.parametrize("M,N,K", [(1024, 512, 128), (1536, 2048, 1024), (700, 528, 64), (256, 64, 64)])
def test_gemm_glu_epilogue(lib, M, N, K):
    """ACT_GLU (round 6): pointwise_conv1 + GLU (convolution.py:107-111) in the GEMM's epilogue -- the weight rows interleaved so that
    output columns 2c / 2c + 1 are the pair (a_c, b_c), stored: a_c * sigmoid(b_c), N / 2 columns.  Against fp64 on full tiles (inline-asm
    8-byte stores) and on ragged ones (the generic loop: 700 rows, 528 columns)."""
    rng = np.random.default_rng(M + N)
    A = rnd(BF16, rng.standard_normal((M, K)))
    W = rnd(BF16, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    C = np.full((M, N // 2), np.nan, np.float32)
    _lib.check(lib.rvb_test_gemm_glu(fptr(A), fptr(W), fptr(bias), fptr(C), M, N, K))
    v = A.astype(np.float64) @ W.astype(np.float64).T + bias
    ref = v[:, 0::2] / (1.0 + np.exp(-v[:, 1::2]))
    np.testing.assert_allclose(C, ref, rtol=1e-2, atol=1e-2)          # one bf16 rounding of the output, hardware exp2 / rcp in the gate



@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("Cc,N", [(16, 24), (64, 72), (128, 260)])
def test_gemm_implicit_conv(lib, dtype, Cc, N):
    """second Conv2d(d,d,3,2)+ReLU of Conv2dSubsampling4 (subsampling.py:189-190) as implicit GEMM on NHWC."""
    B, T1, F1, Cc, N = 2, 21, 15, Cc, N
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    rng = np.random.default_rng(3)
    x = rnd(dtype, rng.standard_normal((B, T1, F1, Cc)))                 # NHWC
    w = rnd(dtype, rng.standard_normal((N, Cc, 3, 3)) / math.sqrt(9 * Cc))  # torch layout [co][ci][kh][kw]
    bias = f32(rng.standard_normal(N))
    wp = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(N, 9 * Cc))   # [co][(kh,kw,ci)]
    M = B * T2 * F2
    C = np.empty((M, N), np.float32)
    _lib.check(lib.rvb_test_gemm(dtype, fptr(x), fptr(wp), fptr(bias), None, fptr(C), M, N, 9 * Cc, 1.0, 2, 1,
                                 1, T1, F1, Cc, B))
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                                                torch.from_numpy(w).double(), torch.from_numpy(bias).double(), stride=2))
    ref = ref.permute(0, 2, 3, 1).reshape(M, N).numpy()                  # (b, t2, f2, co)
    np.testing.assert_allclose(C, ref, rtol=5e-5, atol=1e-4)


@pytest.mark.parametrize("flags", [0])
@pytest.mark.parametrize("Cc,N,B,T1,F1", [(128, 260, 3, 41, 23), (64, 256, 2, 65, 39)])
def test_gemm_implicit_conv_on_the_lds_dma_loops(lib, flags, Cc, N, B, T1, F1):
    """the convolution gather of gemm2p_kernel (A rows addressed through the 3x3 stride-2 taps, 9 Cc / 64 K steps, the kernel-row jump
    of the A offset), several row tiles, ragged edges, bf16 out (the two-phase form of the loop, flag bit 14 with -DRVB_GEMM2_PH2, passed the
    same test in round 6 before it was measured slower and compiled out)"""
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    rng = np.random.default_rng(Cc + N)
    x = rnd(BF16, rng.standard_normal((B, T1, F1, Cc)))
    w = rnd(BF16, rng.standard_normal((N, Cc, 3, 3)) / math.sqrt(9 * Cc))
    bias = f32(rng.standard_normal(N))
    wp = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(N, 9 * Cc))
    M = B * T2 * F2
    C = np.full((M, N), np.nan, np.float32)
    lib.rvb_test_set_gemm2_opts(flags, -2)
    try:
        _lib.check(lib.rvb_test_gemm(BF16, fptr(x), fptr(wp), fptr(bias), None, fptr(C), M, N, 9 * Cc, 1.0, 2, 0, 1, T1, F1, Cc, B))
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                                                torch.from_numpy(w).double(), torch.from_numpy(bias).double(), stride=2))
    ref = ref.permute(0, 2, 3, 1).reshape(M, N).numpy()
    np.testing.assert_allclose(C, ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mode,silu,use_add,out_f32,d", [(0, 0, False, 0, 64), (0, 1, False, 0, 640), (0, 0, True, 1, 1024),
                                                         (1, 1, False, 0, 96)])
def test_rownorm(lib, dtype, mode, silu, use_add, out_f32, d):
    M = 37
    rng = np.random.default_rng(d)
    x = f32(rng.standard_normal((M, d)) * 3 + 1)
    g = f32(1 + 0.1 * rng.standard_normal(d)); b = f32(0.1 * rng.standard_normal(d))
    add = rnd(dtype, rng.standard_normal((M, d))) if use_add else None
    out = np.empty((M, d), np.float32)
    _lib.check(lib.rvb_test_rownorm(dtype, fptr(x), fptr(g), fptr(b), 1e-5, mode, silu, fptr(add), fptr(out), out_f32, M, d))
    xd = x.astype(np.float64)
    if mode == 0:
        mu = xd.mean(1, keepdims=True); var = xd.var(1, keepdims=True)
        ref = (xd - mu) / np.sqrt(var + 1e-5) * g + b
    else:
        ref = xd * g + b
    if silu:
        ref = ref / (1 + np.exp(-ref))
    if use_add:
        ref = ref + add
    tol = 1e-5 if (dtype == F32 or out_f32) else 1e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * 4)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("d", [32, 640])
def test_conv1_cmvn(lib, dtype, d):
    B, T0, F0 = 2, 39, 80
    rng = np.random.default_rng(d)
    feats = f32(rng.standard_normal((B, T0, F0)) * 4 + 15)
    mean = f32(15 + rng.standard_normal(F0)); istd = f32(0.25 + 0.05 * rng.random(F0))
    w = f32(rng.standard_normal((d, 1, 3, 3)) / 3); b = f32(rng.standard_normal(d))
    T1, F1 = (T0 - 3) // 2 + 1, (F0 - 3) // 2 + 1
    out = np.empty((B, T1, F1, d), np.float32)
    _lib.check(lib.rvb_test_conv1(dtype, fptr(feats), fptr(mean), fptr(istd), fptr(w), fptr(b), fptr(out), B, T0, F0, d))
    xn = (torch.from_numpy(feats).double() - torch.from_numpy(mean).double()) * torch.from_numpy(istd).double()
    ref = torch.relu(torch.nn.functional.conv2d(xn.unsqueeze(1), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2))
    ref = ref.permute(0, 2, 3, 1).numpy()
    tol = 2e-5 if dtype == F32 else 1e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * 4)


@pytest.mark.parametrize("flags", [0, 128, 256])
@pytest.mark.parametrize("M,N,K,use_res,out_f32", [(4352, 4400, 1344, False, 0), (4100, 4352, 2112, True, 1)])
def test_gemm2p_ring_many_k_steps_and_reversed_tiles(lib, flags, M, N, K, use_res, out_f32):
    """the phase loop's half-tile ring (LDS slots refilled by LDS-DMA behind counted `vmcnt` waits; WAR safety argued
    in gemm2.hip) under more than a toy load -- 21 and 33 K steps (odd counts: the compile-time tail of round 4 takes its three
    forms on top of an odd number of steady steps), 17 x 18 tiles with ragged edges, and more than 256 tiles so that the K
    serpentine (bit 7 / 8: tiles of the second wave walk K downwards) really reverses some -- against fp64."""
    dtype = BF16
    rng = np.random.default_rng(M + K)
    A = rnd(dtype, rng.standard_normal((M, K)))
    W = rnd(dtype, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    res = f32(rng.standard_normal((M, N))) if use_res else None
    C = np.full((M, N), np.nan, np.float32)
    lib.rvb_test_set_gemm2_opts(flags, -2)
    try:
        _lib.check(lib.rvb_test_gemm(dtype, fptr(A), fptr(W), fptr(bias), fptr(res), fptr(C), M, N, K, 1.0, 0, out_f32, 0, 0, 0, 0, 0))
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)
    v = A.astype(np.float64) @ W.astype(np.float64).T + bias + (res if use_res else 0)
    if out_f32:
        np.testing.assert_allclose(C, v, rtol=5e-5, atol=3e-4)
    else:
        np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("flags", [0, 16, 4096])
@pytest.mark.parametrize("M,N,K,act,alpha,use_res,out_f32,group_m", [
    (8192, 4096, 320, 1, 1.0, False, 0, -2),     # 512 tiles >= 2 per CU: the persistent form (cross-tile prefetch), SiLU, bf16 out
    (16384, 2048, 256, 0, 0.5, True, 1, 8),      # residual preloaded into the accumulators tile after tile, grouped order, nk = 4
    (12288, 3072, 1024, 0, 1.0, False, 0, 0),    # row-major order, 3 tiles per CU (the last round partial)
])
def test_gemm2_persistent_tiles(lib, flags, M, N, K, act, alpha, use_res, out_f32, group_m):
    """The persistent form of the phase-interleaved loop (flags 16, opt-in: one workgroup per CU walks the tiles of its XCD's
    run, the ring keeps streaming across tile edges) against fp64 -- every tile of a large problem, and bit-identical to one
    tile per workgroup (flags 0), which runs the same arithmetic in the same order.  flags 4096: the persistent LOOP without the
    cross-tile prefetch (only the dispatch gap between two workgroups of a CU goes)."""
    rng = np.random.default_rng(M + N + K)
    A = rnd(BF16, rng.standard_normal((M, K)))
    W = rnd(BF16, rng.standard_normal((N, K)) / math.sqrt(K))
    bias = f32(rng.standard_normal(N))
    res = f32(rng.standard_normal((M, N))) if use_res else None
    C = np.full((M, N), np.nan, np.float32)
    lib.rvb_test_set_gemm2_opts(flags, group_m)
    try:
        _lib.check(lib.rvb_test_gemm(BF16, fptr(A), fptr(W), fptr(bias), fptr(res), fptr(C), M, N, K, alpha, act, out_f32,
                                     0, 0, 0, 0, 0))
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)
    v = A.astype(np.float32) @ W.astype(np.float32).T           # bf16 products are exact in fp32; the sum order differs
    v = v.astype(np.float64) + bias
    if act == 1:
        v = v / (1 + np.exp(-v))
    v = v * alpha + (res if use_res else 0)
    assert np.isfinite(C).all()
    if out_f32:
        np.testing.assert_allclose(C, v, rtol=2e-4, atol=1e-3)
    else:
        np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)
    key = (M, N, K)
    if flags == 0:
        _PERSIST_RESULTS[key] = C.copy()
    elif key in _PERSIST_RESULTS:
        np.testing.assert_array_equal(C, _PERSIST_RESULTS[key])


_PERSIST_RESULTS = {}


def _assert_glu_dwconv(out, ref, glu, w, dtype, padding):
    """f32 engine: fp32 arithmetic throughout.  bf16 engine: the gated value is kept in bf16 (as F.glu under autocast returns
    it), so every tap carries one bf16 rounding of its input (half an ulp of an 8-bit significand): |error| <= sum_k |w_k| |glu_k| 2^-8
    (+ fp32 noise)."""
    if dtype == F32:
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=1e-4)
        return
    wd = torch.from_numpy(np.abs(w)).double().unsqueeze(1)
    bound = torch.nn.functional.conv1d(glu.abs(), wd, None, padding=padding, groups=w.shape[0]).transpose(1, 2).numpy() * 2.0 ** -8
    err = np.abs(out - ref)
    assert np.all(err <= bound * 1.01 + 1e-4), float((err - bound).max())
    assert err.mean() < 0.4 * bound.mean() + 1e-5           # roundings are unbiased: well inside the worst case on average


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("K,d,T", [(15, 32, 70), (31, 200, 130), (31, 128, 300)])
def test_glu_dwconv(lib, dtype, K, d, T):
    B = 3
    rng = np.random.default_rng(K)
    lens = i32([T, T - 9, 5])
    G = rnd(dtype, rng.standard_normal((B, T, 2 * d)))
    pb = f32(rng.standard_normal(2 * d)); w = f32(rng.standard_normal((d, K)) / math.sqrt(K)); b = f32(rng.standard_normal(d))
    out = np.empty((B, T, d), np.float32)
    _lib.check(lib.rvb_test_glu_dwconv(dtype, fptr(G), fptr(pb), fptr(w), fptr(b), iptr(lens), fptr(out), B, T, d, K, 0, None, 0))
    # reference semantics (convolution.py:107-131): padded frames were zeroed BEFORE pointwise_conv1,
    # so what the GLU sees there is the bias alone
    Gd = torch.from_numpy(G).double().clone()
    for bi in range(B):
        Gd[bi, lens[bi]:, :] = torch.from_numpy(pb).double()
    glu = torch.nn.functional.glu(Gd.transpose(1, 2), dim=1)
    ref = torch.nn.functional.conv1d(glu, torch.from_numpy(w).double().unsqueeze(1), torch.from_numpy(b).double(),
                                     padding=(K - 1) // 2, groups=d).transpose(1, 2).numpy()
    _assert_glu_dwconv(out, ref, glu, w, dtype, (K - 1) // 2)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("K,d,T,hist_rows", [(15, 32, 70, 0), (8, 200, 130, 7), (31, 64, 5, 11), (15, 32, 3, 14), (2, 16, 9, 1)])
def test_glu_dwconv_causal(lib, dtype, K, d, T, hist_rows):
    """Causal convolution module (convolution.py:55-57,113-125): K-1 frames of left context -- cached frames (the last
    hist_rows of the K-1) or the zero padding seen through pointwise_conv1 + GLU; padded batch rows see the bias too."""
    rng = np.random.default_rng(K * 100 + T)
    B = 1 if hist_rows else 3
    lens = i32([T] if hist_rows else [T, max(T - 9, 1), min(5, T)])
    G = rnd(dtype, rng.standard_normal((B, T, 2 * d)))
    hist = rnd(dtype, rng.standard_normal((K - 1, 2 * d)))
    pb = f32(rng.standard_normal(2 * d)); w = f32(rng.standard_normal((d, K)) / math.sqrt(K)); b = f32(rng.standard_normal(d))
    out = np.empty((B, T, d), np.float32)
    _lib.check(lib.rvb_test_glu_dwconv(dtype, fptr(G), fptr(pb), fptr(w), fptr(b), iptr(lens), fptr(out), B, T, d, K, 1,
                                       fptr(hist) if hist_rows else None, hist_rows))
    Gd = torch.from_numpy(G).double().clone()
    for bi in range(B):
        Gd[bi, lens[bi]:, :] = torch.from_numpy(pb).double()
    left = torch.from_numpy(pb).double().repeat(B, K - 1, 1)          # pointwise_conv1 of a zero frame = its bias
    if hist_rows:
        left[0, K - 1 - hist_rows:] = torch.from_numpy(hist).double()[K - 1 - hist_rows:]
    glu = torch.nn.functional.glu(torch.cat([left, Gd], 1).transpose(1, 2), dim=1)
    ref = torch.nn.functional.conv1d(glu, torch.from_numpy(w).double().unsqueeze(1), torch.from_numpy(b).double(),
                                     groups=d).transpose(1, 2).numpy()
    assert ref.shape == out.shape
    _assert_glu_dwconv(out, ref, glu, w, dtype, 0)


def test_bf16_engine_dwconv_output_and_norm_input(lib):
    """bf16 engine: the depthwise convolution writes bf16 and the convolution-module norm reads bf16 (half the bytes of the
    fp32 round trip) -- the same values up to that one rounding."""
    rng = np.random.default_rng(5)
    B, T, d, K = 2, 70, 64, 15
    lens = i32([T, T - 9])
    G = rnd(BF16, rng.standard_normal((B, T, 2 * d)))
    pb = f32(rng.standard_normal(2 * d)); w = f32(rng.standard_normal((d, K)) / math.sqrt(K)); b = f32(rng.standard_normal(d))
    o32 = np.empty((B, T, d), np.float32); o16 = np.empty((B, T, d), np.float32)
    _lib.check(lib.rvb_test_glu_dwconv(BF16, fptr(G), fptr(pb), fptr(w), fptr(b), iptr(lens), fptr(o32), B, T, d, K, 0, None, 0))
    _lib.check(lib.rvb_test_glu_dwconv(BF16, fptr(G), fptr(pb), fptr(w), fptr(b), iptr(lens), fptr(o16), B, T, d, K, 2, None, 0))
    np.testing.assert_array_equal(o16, bf16_round(o32))
    M, d = 37, 256
    x = bf16_round(rng.standard_normal((M, d)) * 3 + 1)
    g = f32(1 + 0.1 * rng.standard_normal(d)); be = f32(0.1 * rng.standard_normal(d))
    a = np.empty((M, d), np.float32); c = np.empty((M, d), np.float32)
    _lib.check(lib.rvb_test_rownorm(BF16, fptr(x), fptr(g), fptr(be), 1e-5, 0, 1, None, fptr(a), 0, M, d))
    _lib.check(lib.rvb_test_rownorm(BF16, fptr(x), fptr(g), fptr(be), 1e-5, 256, 1, None, fptr(c), 0, M, d))
    np.testing.assert_array_equal(a, c)          # x is exactly representable in bf16: both input forms give the same output


def _ref_attention(q, k, v, p, bu, bv, heads, dk, q_start, q_len, kv_start, kv_len, causal):
    d = heads * dk
    out = np.zeros((q.shape[0], d))
    for s in range(len(q_start)):
        qs, ql, ks, kl = int(q_start[s]), int(q_len[s]), int(kv_start[s]), int(kv_len[s])
        for h in range(heads):
            sl = slice(h * dk, (h + 1) * dk)
            Q = q[qs:qs + ql, sl].astype(np.float64)
            Kk = k[ks:ks + kl, sl].astype(np.float64); V = v[ks:ks + kl, sl].astype(np.float64)
            if kl == 0:
                continue
            if p is not None:
                S = (Q + bu[sl]) @ Kk.T + (Q + bv[sl]) @ p[:kl, sl].astype(np.float64).T
            else:
                S = Q @ Kk.T
            S = S / math.sqrt(dk)
            if causal:
                S = np.where(np.arange(kl)[None, :] > np.arange(ql)[:, None], -np.inf, S)
            S = S - S.max(1, keepdims=True)
            P = np.exp(S); P /= P.sum(1, keepdims=True)
            out[qs:qs + ql, sl] = P @ V
    return out


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("heads,dk,T,pos", [(2, 16, 100, True), (4, 64, 200, True), (2, 64, 517, True), (3, 48, 129, True), (2, 80, 130, True), (2, 32, 70, False)])
def test_attention_encoder_form(lib, dtype, heads, dk, T, pos):
    """RelPositionMultiHeadedAttention scores (attention.py:378-399) with key-padding mask."""
    B = 3
    d = heads * dk
    rng = np.random.default_rng(heads * 1000 + dk)
    kv_len = i32([T, T - 37, 0])
    q = rnd(dtype, rng.standard_normal((B * T, d)))
    k = rnd(dtype, rng.standard_normal((B * T, d)))
    v = rnd(dtype, rng.standard_normal((B * T, d)))
    p = rnd(dtype, rng.standard_normal((T, d))) if pos else None
    bu = f32(0.3 * rng.standard_normal(d)) if pos else None
    bv = f32(0.3 * rng.standard_normal(d)) if pos else None
    starts = i32([0, T, 2 * T]); qlen = i32([T, T, T])
    out = np.empty((B * T, d), np.float32)
    _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), fptr(p), fptr(bu), fptr(bv), fptr(out), B * T, B * T,
                                      T if pos else 0, heads, dk, iptr(starts), iptr(qlen), iptr(starts), iptr(kv_len), B, 0))
    ref = _ref_attention(q, k, v, p, bu, bv, heads, dk, starts, qlen, starts, kv_len, False)
    tol = 2e-5 if dtype == F32 else 3e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)
    assert np.all(out[2 * T:] == 0)     # fully masked chunk -> zeros (attention.py:112-114)
    if pos and dtype == BF16 and 32 < dk <= 64:
        # round 4: the form the bf16 engine runs -- (q + u).(k + p) + (v - u).p, the second product a per-key table (attention.hip
        # FOLD; bit 1 of the test hook's `causal` word builds the table with attention_pos_bias): same scores up to bf16 rounding
        # of k + p, so the same tolerance against the fp64 reference
        out2 = np.empty_like(out)
        _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), fptr(p), fptr(bu), fptr(bv), fptr(out2), B * T, B * T, T,
                                          heads, dk, iptr(starts), iptr(qlen), iptr(starts), iptr(kv_len), B, 2))
        np.testing.assert_allclose(out2, ref, rtol=tol, atol=tol)
        assert np.all(out2[2 * T:] == 0)
        assert not np.array_equal(out2, out)          # it really is the other kernel
        # round 6: the same with K' = k + p handed over ready-made (bit 2: what the qkv GEMM's epilogue writes in the engine); from 129
        # queries on this is the kernel with two 16-query fragments per wave and the per-key constants resident in LDS
        out3 = np.empty_like(out)
        _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), fptr(p), fptr(bu), fptr(bv), fptr(out3), B * T, B * T, T,
                                          heads, dk, iptr(starts), iptr(qlen), iptr(starts), iptr(kv_len), B, 6))
        np.testing.assert_allclose(out3, ref, rtol=tol, atol=tol)
        assert np.all(out3[2 * T:] == 0)
        # the fold while staging forms the same K' (fp32 sum of the two bf16 values, one rounding) and a query's arithmetic does not
        # depend on how many fragments its wave owns: the two kernels agree bit for bit
        assert np.array_equal(out3, out2)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attention_late_maximum_forces_a_rescale(lib, dtype):
    """bf16 mode defers the rescaling of the online softmax while no query's maximum grows by more than 2^8: random scores
    never take the rescale branch after the first tile, so keys are spiked against chosen queries in LATER tiles (tile 2 for
    some queries, tile 5 for others, both for a third group, and a small growth below the threshold for a fourth) -- every
    row against fp64."""
    heads, dk, T = 2, 64, 400
    d = heads * dk
    rng = np.random.default_rng(77)
    q = rng.standard_normal((T, d)); k = rng.standard_normal((T, d)); v = rng.standard_normal((T, d))
    for h in range(heads):
        sl = slice(h * dk, (h + 1) * dk)
        # group A (queries 0, 3, 6, ..): key 150 (tile 2) lies along their common direction, ~20 nats above their other
        # scores; group B (queries 1, 4, ..): the same with key 330 (tile 5)
        qa = q[0:T:3, sl].mean(0); ua = qa / np.linalg.norm(qa)
        qb = q[1:T:3, sl].mean(0); ub = qb / np.linalg.norm(qb)
        k[150, sl] = ua * 6.0 * math.sqrt(dk) / np.linalg.norm(qa)
        k[330, sl] = ub * 6.0 * math.sqrt(dk) / np.linalg.norm(qb)
        q[0:T:3, sl] += 2.0 * ua
        q[1:T:3, sl] += 2.0 * ub
        k[200, sl] = 0.4 * k[200, sl] + 0.3 * q[2, sl]      # group C: a mild growth that stays under the threshold
    q, k, v = rnd(dtype, q), rnd(dtype, k), rnd(dtype, v)
    starts = i32([0]); lens = i32([T])
    out = np.empty((T, d), np.float32)
    _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), None, None, None, fptr(out), T, T, 0, heads, dk,
                                      iptr(starts), iptr(lens), iptr(starts), iptr(lens), 1, 0))
    ref = _ref_attention(q, k, v, None, None, None, heads, dk, starts, lens, starts, lens, False)
    # the spikes really are late maxima far above the threshold (8 in the log2 domain = 5.5 nats)
    sc = q[:, :dk].astype(np.float64) @ k[:, :dk].astype(np.float64).T / math.sqrt(dk)
    assert (sc[0::3, 150] - np.delete(sc[0::3], [150, 330], axis=1)[:, :128].max(1) > 6).mean() > 0.9
    tol = 2e-5 if dtype == F32 else 3e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attention_decoder_forms(lib, dtype):
    """causal ragged self attention and cross attention over a chunk memory (decoder.py:150-156)."""
    heads, dk = 4, 32
    d = heads * dk
    rng = np.random.default_rng(11)
    qlen = i32([5, 83, 1, 130]); qstart = i32(np.concatenate([[0], np.cumsum(qlen)[:-1]]))
    R = int(qlen.sum())
    q = rnd(dtype, rng.standard_normal((R, d))); k = rnd(dtype, rng.standard_normal((R, d))); v = rnd(dtype, rng.standard_normal((R, d)))
    out = np.empty((R, d), np.float32)
    _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), None, None, None, fptr(out), R, R, 0, heads, dk,
                                      iptr(qstart), iptr(qlen), iptr(qstart), iptr(qlen), 4, 1))
    ref = _ref_attention(q, k, v, None, None, None, heads, dk, qstart, qlen, qstart, qlen, True)
    tol = 2e-5 if dtype == F32 else 3e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)
    # cross attention: 2 chunks of 90 memory frames, valid 90 / 41
    Tm = 90
    mem_k = rnd(dtype, rng.standard_normal((2 * Tm, d))); mem_v = rnd(dtype, rng.standard_normal((2 * Tm, d)))
    kvs = i32([0, 0, Tm, Tm]); kvl = i32([90, 90, 41, 41])
    _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(mem_k), fptr(mem_v), None, None, None, fptr(out), R, 2 * Tm, 0,
                                      heads, dk, iptr(qstart), iptr(qlen), iptr(kvs), iptr(kvl), 4, 0))
    ref = _ref_attention(q, mem_k, mem_v, None, None, None, heads, dk, qstart, qlen, kvs, kvl, False)
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("q_block", [16, 0])
def test_attention_over_a_hypothesis_trie(lib, dtype, q_block):
    """Rescoring computes one decoder row per DISTINCT hypothesis prefix: a hypothesis' queries are the rows it adds to
    the trie (positions q_pos0..), its keys the rows of its whole prefix path (an index list).  Reference: plain causal
    attention of every hypothesis over its own gathered rows."""
    heads, dk = 4, 32
    d = heads * dk
    rng = np.random.default_rng(3 + q_block)
    # a trie of 5 hypotheses: rows in creation order; paths share prefixes of different lengths
    paths = [list(range(0, 90)),                                   # hyp 0 owns rows 0..89 (positions 0..89)
             list(range(0, 70)) + list(range(90, 110)),            # shares 70, owns 20 (positions 70..89)
             list(range(0, 70)) + list(range(90, 95)) + [110],     # shares 75 with hyp 1, owns 1 row
             list(range(0, 3)),                                    # a strict prefix of hyp 0: owns nothing
             list(range(0, 1)) + list(range(111, 260))]            # shares only <sos>, owns 149 (ten 16-query blocks)
    rows = 260
    q, k, v = (rnd(dtype, rng.standard_normal((rows, d))) for _ in range(3))
    q_start, q_len, q_pos0, kv_start, kv_len, index = [], [], [], [], [], []
    seen = set()
    for p in paths:
        own = [r for r in p if r not in seen]
        seen.update(p)
        kv_start.append(len(index)); kv_len.append(len(p)); index += p
        q_start.append(own[0] if own else 0); q_len.append(len(own)); q_pos0.append(p.index(own[0]) if own else 0)
        assert own == list(range(q_start[-1], q_start[-1] + len(own))) and own == p[len(p) - len(own):]
    out = np.empty((rows, d), np.float32)
    _lib.check(lib.rvb_test_attention_trie(dtype, fptr(q), fptr(k), fptr(v), fptr(out), rows, heads, dk, iptr(i32(q_start)),
                                           iptr(i32(q_len)), iptr(i32(q_pos0)), iptr(i32(kv_start)), iptr(i32(kv_len)),
                                           iptr(i32(index)), len(index), len(paths), q_block))
    ref = np.zeros((rows, d))
    for p in paths:
        idx = np.array(p)
        L = len(p)
        full = _ref_attention(q[idx], k[idx], v[idx], None, None, None, heads, dk, i32([0]), i32([L]), i32([0]), i32([L]), True)
        ref[idx] = full                      # shared rows are written several times with the same values
    tol = 2e-5 if dtype == F32 else 3e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("V", [1001, 1000, 10004])
def test_logsoftmax_topk(lib, V):
    M, k = 37, 10
    rng = np.random.default_rng(5)
    x = f32(rng.standard_normal((M, V)) * 3)
    x[3, 17] = x[3, 400] = x[3].max() + 1.0          # exact tie: lower index first
    x[5, :] = 0.25                                    # all-equal row: > 256 candidates -> exclusion-scan fallback
    x[6, :] = 0.5; x[6, 700:] = 0.75                  # 300-way tie at the top
    tv = np.empty((M, k), np.float32); ti = np.empty((M, k), np.int32); lp = np.empty((M, V), np.float32)
    _lib.check(lib.rvb_test_logsoftmax_topk(fptr(x), M, V, k, 0.0, 0, fptr(tv), iptr(ti), fptr(lp)))
    ref = torch.from_numpy(x).double().log_softmax(-1)
    np.testing.assert_allclose(lp, ref.numpy(), rtol=0, atol=2e-6)
    rv, ri = ref.float().topk(k, dim=-1)
    assert ti[3, 0] == 17 and ti[3, 1] == 400
    assert ti[5].tolist() == list(range(k))                 # ties -> ascending index (stable descending order)
    assert ti[6].tolist() == list(range(700, 700 + k))
    rows = [r for r in range(M) if r not in (3, 5, 6)]      # torch's own tie order is unspecified
    np.testing.assert_array_equal(ti[rows], ri.numpy()[rows])
    np.testing.assert_allclose(tv, rv.numpy(), rtol=0, atol=2e-6)
    # blank penalty (asr_model.py:322-325)
    _lib.check(lib.rvb_test_logsoftmax_topk(fptr(x), M, V, k, 2.5, 0, fptr(tv), iptr(ti), fptr(lp)))
    x2 = x.copy(); x2[:, 0] -= 2.5
    np.testing.assert_allclose(lp, torch.from_numpy(x2).double().log_softmax(-1).numpy(), rtol=0, atol=2e-6)


def test_lse_gather_multi_targets_per_row(lib):
    """CSR form used by the trie rescoring: a row shared by several hypotheses is asked for several targets."""
    V, R = 10001, 9
    rng = np.random.default_rng(8)
    x = f32(rng.standard_normal((R, V)) * 2)
    counts = [1, 3, 0, 70, 2, 1, 1, 5, 2]            # a row without asks, a row with more asks than lanes
    ptr = i32(np.concatenate([[0], np.cumsum(counts)]))
    tgt = i32(rng.integers(0, V, int(ptr[-1])))
    out = np.full(int(ptr[-1]), np.nan, np.float32)
    _lib.check(lib.rvb_test_lse_gather_multi(fptr(x), R, V, iptr(ptr), iptr(tgt), int(ptr[-1]), fptr(out)))
    ref = torch.from_numpy(x).double().log_softmax(-1).numpy()
    want = np.concatenate([ref[r, tgt[ptr[r]:ptr[r + 1]]] for r in range(R)])
    np.testing.assert_allclose(out, want, rtol=0, atol=3e-6)


@pytest.mark.parametrize("V", [10001, 10004])
def test_lse_gather(lib, V):
    R = 29
    rng = np.random.default_rng(9)
    x = f32(rng.standard_normal((R, V)) * 2)
    tgt = i32(rng.integers(0, V, R))
    out = np.empty(R, np.float32)
    _lib.check(lib.rvb_test_lse_gather(fptr(x), R, V, iptr(tgt), fptr(out)))
    ref = torch.from_numpy(x).double().log_softmax(-1).numpy()[np.arange(R), tgt]
    np.testing.assert_allclose(out, ref, rtol=0, atol=3e-6)


def test_fbank_vs_oracle(lib):
    from oracle import fbank_ref
    from reverb_amd import synth
    pcm = synth.synth_audio(3.3, seed=7)
    nf = fbank_ref.num_frames(len(pcm))
    feats = np.empty((nf, 80), np.float32)
    _lib.check(lib.rvb_test_fbank(pcm.ctypes.data_as(_lib._i16p), len(pcm), fptr(feats)))
    ref = fbank_ref.fbank(pcm)
    assert ref.shape == feats.shape
    # fp32 radix-2 FFT vs pocketfft: log-mel agrees to 1e-3 abs (SURVEY.md 8d fbank tolerance)
    np.testing.assert_allclose(feats, ref, rtol=0, atol=1e-3)
    # low-amplitude input: every bin hits the log floor path or tiny energies
    quiet = (pcm // 4000).astype(np.int16)
    _lib.check(lib.rvb_test_fbank(quiet.ctypes.data_as(_lib._i16p), len(quiet), fptr(feats)))
    np.testing.assert_allclose(feats, fbank_ref.fbank(quiet), rtol=0, atol=2e-3)


@pytest.mark.parametrize("bins,flen,fshift", [(23, 25, 10), (40, 25, 10), (80, 32, 8), (128, 20, 5), (64, 30, 12.5), (80, 25, 10)])
def test_compute_feats_with_other_front_end_settings(bins, flen, fshift):
    """rvb_compute_feats (VERDICT r4 "missing" #7): `ReverbASR.compute_feats` passes any num_mel_bins / frame_length / frame_shift
    to kaldi.fbank (cli/reverb.py:119-146; its own default is 23 bins).  The device kernel with those sizes against the oracle's
    restatement of torchaudio's fbank, same tolerance as the hot-path kernel, incl. the model's own setting."""
    import ctypes as C
    from oracle import fbank_ref
    from reverb_amd import synth
    product = _lib.load()
    pcm = synth.synth_audio(2.7, seed=11)
    wave = pcm.astype(np.float32)
    n = C.c_int64(0)
    _lib.check(product.rvb_compute_feats(0, fptr(wave), wave.size, bins, float(flen), float(fshift), None, C.byref(n)))
    ref = fbank_ref.fbank(pcm, bins, flen, fshift)
    assert n.value == ref.shape[0] > 100
    got = np.empty((n.value, bins), np.float32)
    _lib.check(product.rvb_compute_feats(0, fptr(wave), wave.size, bins, float(flen), float(fshift), fptr(got), C.byref(n)))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-3)
    short = np.zeros(100, np.float32)                               # shorter than one window: no frame
    _lib.check(product.rvb_compute_feats(0, fptr(short), short.size, bins, float(flen), float(fshift), None, C.byref(n)))
    assert n.value == 0


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("chunk,left", [(16, -1), (16, 2), (7, 1), (64, 0), (200, 3)])
def test_attention_chunk_mask(lib, dtype, chunk, left):
    """subsequent_chunk_mask (utils/mask.py:86-123) inside the kernel: query i sees keys
    [max((i/cs - left)*cs, 0), (i/cs + 1)*cs) of its own sequence; tiles outside the range are skipped."""
    rng = np.random.default_rng(chunk * 31 + left + dtype)
    lens = i32([300, 77, 1])
    heads, dk = 2, 64
    d = heads * dk
    starts = i32(np.concatenate([[0], np.cumsum(lens)[:-1]]))
    rows = int(lens.sum())
    q, k, v = (rnd(dtype, rng.standard_normal((rows, d))) for _ in range(3))
    out = np.empty((rows, d), np.float32)
    code = ((left + 1) << 20) | (chunk << 8)          # test-API packing of (chunk, left) into the `causal` argument
    _lib.check(lib.rvb_test_attention(dtype, fptr(q), fptr(k), fptr(v), None, None, None, fptr(out), rows, rows, 0, heads, dk,
                                      iptr(starts), iptr(lens), iptr(starts), iptr(lens), 3, code))
    ref = np.zeros((rows, d))
    for s0, L in zip(starts, lens):
        idx = np.arange(L)
        lo = np.zeros(L, int) if left < 0 else np.maximum((idx // chunk - left) * chunk, 0)
        hi = np.minimum((idx // chunk + 1) * chunk, L)
        mask = (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])
        for h in range(heads):
            sl = slice(h * dk, (h + 1) * dk)
            sc = q[s0:s0 + L, sl].astype(np.float64) @ k[s0:s0 + L, sl].astype(np.float64).T / math.sqrt(dk)
            sc = np.where(mask, sc, -np.inf)
            pr = np.exp(sc - sc.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
            ref[s0:s0 + L, sl] = pr @ v[s0:s0 + L, sl].astype(np.float64)
    tol = 2e-5 if dtype == F32 else 3e-2
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol)
