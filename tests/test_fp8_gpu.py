"""RVB_FP8 mode (BASELINE configs[4] "fp8 MFMA GEMMs"): the fp8 (OCP e4m3) GEMM of gemm2.hip against fp64 on exactly the
values its quantised operands stand for (so the check isolates the kernel: operand layout of
v_mfma_scale_f32_32x32x64_f8f6f4, per-channel / per-tensor scales, every epilogue form), the LayerNorm-to-fp8 kernels, and
the engine end to end against the reference goldens: token error rate with the measured bound written below."""
import math

import numpy as np
import pytest

from reverb_amd import _lib
from reverb_amd._lib import fptr
from util import f32

pytestmark = pytest.mark.gpu


def _e4m3_grid():
    v = [0.0]
    for E in range(16):
        for M in range(8):
            if E == 15 and M == 7:
                continue
            v.append((M / 8.0) * 2.0 ** -6 if E == 0 else (1 + M / 8.0) * 2.0 ** (E - 7))
    g = np.array(sorted(set(v)))
    return np.concatenate([-g[::-1], g])


@pytest.mark.parametrize("M,N,K,act,alpha,use_res,out_kind", [
    (300, 256, 128, 0, 1.0, False, 1),          # one K step, a partial row tile
    (1000, 1024, 1024, 1, 1.0, False, 2),       # SiLU, fp8 output (the feed-forward's h)
    (513, 3072, 256, 0, 1.0, False, 0),         # bf16 output (qkv)
    (2100, 512, 512, 0, 0.5, True, 1),          # fp32 residual epilogue (feed-forward's second GEMM)
    (130, 330, 384, 2, 1.0, False, 1),          # unaligned rows: element-wise epilogue
])
@pytest.mark.parametrize("flags", [0, 8])        # 8: the phase-interleaved loop (opt-in for fp8: measured slower on K = 1024)
def test_fp8_gemm_against_fp64_on_the_quantised_values(lib, flags, M, N, K, act, alpha, use_res, out_kind):
    lib.rvb_test_set_gemm2_opts(flags, -2)
    try:
        _fp8_gemm_case(lib, M, N, K, act, alpha, use_res, out_kind)
    finally:
        lib.rvb_test_set_gemm2_opts(-1, -1)


def _fp8_gemm_case(lib, M, N, K, act, alpha, use_res, out_kind):
    rng = np.random.default_rng(M + N + K)
    A = f32(rng.standard_normal((M, K)) * 1.7)
    W = f32(rng.standard_normal((N, K)) / math.sqrt(K) * (1 + rng.random((N, 1)) * 3))       # channels of different magnitude
    bias = f32(rng.standard_normal(N))
    res = f32(rng.standard_normal((M, N))) if use_res else None
    a_scale = float(np.abs(A).max() * 2 / 448)
    C = np.full((M, N), np.nan, np.float32)
    Ad, Wd = np.empty_like(A), np.empty_like(W)
    out_scale = 1.0
    v = None
    for attempt in range(2):          # the fp8 output scale comes from the result itself (as calibration does)
        _lib.check(lib.rvb_test_gemm_fp8(fptr(A), fptr(W), fptr(bias), fptr(res), fptr(C), M, N, K, a_scale, alpha, act, out_kind,
                                         out_scale, fptr(Ad), fptr(Wd)))
        v = Ad.astype(np.float64) @ Wd.astype(np.float64).T + bias
        if act == 1:
            v = v / (1 + np.exp(-v))
        elif act == 2:
            v = np.maximum(v, 0)
        v = v * alpha + (res if use_res else 0)
        if out_kind != 2 or attempt == 1:
            break
        out_scale = float(np.abs(v).max() * 2 / 448)
    # the operands really are e4m3 values at the stated scales
    grid = _e4m3_grid()
    q = Ad[:8].astype(np.float64) / a_scale
    nearest = grid[np.abs(q[..., None] - grid).argmin(-1)]
    assert np.abs(q - nearest).max() < 1e-4 * 448
    assert np.abs(Ad - A).max() <= np.abs(A).max() / 16 + 1e-6 and np.abs(Wd - W).max() <= np.abs(W).max() / 16 + 1e-6
    if out_kind == 1:
        # the products of e4m3 values are exact, the sum over K is not an exact fp32 chain: the scaled MFMA aligns the 64
        # products of an instruction before adding them (measured: 2e-4 relative on these operands; small integers are exact,
        # scripts/micro/f8_probe.hip)
        np.testing.assert_allclose(C, v, rtol=2e-3, atol=2e-3)
    elif out_kind == 0:
        np.testing.assert_allclose(C, v, rtol=1e-2, atol=1e-2)                                   # one bf16 rounding
    else:
        np.testing.assert_allclose(C, v, rtol=0.07, atol=out_scale * 2.0 ** -9 * 1.01)           # one e4m3 rounding (3-bit mantissa)


def test_layernorm_to_fp8(lib):
    M, d = 37, 1024
    rng = np.random.default_rng(4)
    x = f32(rng.standard_normal((M, d)) * 3 + 1)
    g, b = f32(1 + 0.1 * rng.standard_normal(d)), f32(0.1 * rng.standard_normal(d))
    g2, b2 = f32(1 + 0.1 * rng.standard_normal(d)), f32(0.1 * rng.standard_normal(d))
    xd = x.astype(np.float64)
    ln = (xd - xd.mean(1, keepdims=True)) / np.sqrt(xd.var(1, keepdims=True) + 1e-5) * g + b
    scale = float(np.abs(ln).max() * 2 / 448)
    out = np.empty((M, d), np.float32)
    _lib.check(lib.rvb_test_rownorm_fp8(fptr(x), fptr(g), fptr(b), 1e-5, 0, M, d, scale, fptr(out), None, None, 0.0, 1.0, None, None))
    np.testing.assert_allclose(out, ln, rtol=0.07, atol=scale * 2.0 ** -9 * 1.01)
    # fused pair: stage 1 fp32 (norm_final), stage 2 = the next block's LayerNorm as fp8
    ln2 = (ln - ln.mean(1, keepdims=True)) / np.sqrt(ln.var(1, keepdims=True) + 1e-5) * g2 + b2
    scale2 = float(np.abs(ln2).max() * 2 / 448)
    o1, o2 = np.empty((M, d), np.float32), np.empty((M, d), np.float32)
    _lib.check(lib.rvb_test_rownorm_fp8(fptr(x), fptr(g), fptr(b), 1e-5, 0, M, d, 1.0, None, fptr(g2), fptr(b2), 1e-5, scale2,
                                        fptr(o1), fptr(o2)))
    np.testing.assert_allclose(o1, ln, rtol=1e-5, atol=4e-5)
    np.testing.assert_allclose(o2, ln2, rtol=0.07, atol=scale2 * 2.0 ** -9 * 1.01)


# fp8 is held to the reference's own reduced-precision behaviour (test_longform_gpu.RefBf16: the unmodified reference under
# torch.autocast(cpu, bfloat16) against its fp32 run), with a slack of 1.5: token error rate <= 1.5 x (reference-bf16 TER +
# margin), frame decisions on confident frames likewise.  Measured round 3 on the bench hour (profiles/r03_fp8_policy_sweep.txt):
# default policy (feed-forward GEMMs in fp8) 9.3 % greedy / 10.9 % rescored, reference bf16 8.9 % / 9.1 %, engine bf16 4.7 % /
# 8.9 %; all five GEMM groups in fp8 (round 2's mode, still selectable) 16.2 % / 13.5 %.  Random-weight models have tiny CTC
# margins, so these bound what calibrated fp8 costs a trained model from above.
FP8_SLACK = 1.5


@pytest.mark.parametrize("name", ["small_66", "r640_chunk"])
def test_fp8_engine_against_reference(name):
    from golden_util import LongCase
    from reverb_amd.engine import Engine
    from test_longform_gpu import MODES, RefBf16, _assert_reduced_precision, _record, _tap_metrics, _ter
    case = LongCase(name)
    n = len(case.js["lens"]) if name == "small_66" else 2
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(n)])
    lens = np.array(case.js["lens"][:n], np.int32)
    eng = Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(x, lens, case.beam)                      # calibration batch: runs in bf16, records the activation ranges
    cal = _ter(eng.search(MODES, case.ctc_weight, case.reverse_weight), case)
    eng.reset_timings()
    eng.encode(x, lens, case.beam)                      # fp8 GEMMs
    assert eng.timing("gemm_fp8")["launches"] > 0, "the fp8 GEMM path did not run"
    m0 = _tap_metrics(eng, case, 0, 0)
    ref = RefBf16(name)
    fm = ref.frame_disagreement(eng)
    ter = _ter(eng.search(MODES, case.ctc_weight, case.reverse_weight), case)
    _record(case=name, dtype="fp8", ter={m: list(v) for m, v in ter.items()}, bf16_calibration_pass_ter={m: list(v) for m, v in cal.items()},
            chunk0=m0, frames=fm, reference_bf16_ter=ref.ter)
    assert m0["cos"] > 0.995, m0
    _assert_reduced_precision(name, "fp8", ter, fm, ref, slack=FP8_SLACK)
    eng.close()


def test_fp8_subsampling_conv2_policy_bit():
    """Round 4: policy bit 5 ("subsample_conv2") -- conv1 writes its ReLU output in e4m3 at a calibrated scale and conv2 (K = 9 d, a
    quarter of the encoder's FLOPs) runs on the fp8 phase loop with the convolution gather.  The bit moves that GEMM's FLOPs from
    the bf16 to the fp8 counter, nothing of conv1's output is clipped on the calibration batch, the encoder output stays the
    reference's (cos), and the token errors are held to the same yardstick as the default policy."""
    from golden_util import LongCase
    from reverb_amd.engine import Engine
    from test_longform_gpu import MODES, RefBf16, _assert_reduced_precision, _record, _tap_metrics, _ter
    name = "r640_chunk"
    case = LongCase(name)
    n = 2
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(n)])
    lens = np.array(case.js["lens"][:n], np.int32)
    eng = Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    assert eng.fp8_subsample() == (0.0, 0)                          # not calibrated yet
    eng.encode(x, lens, case.beam)                                  # calibration batch (bf16)
    scale, clipped = eng.fp8_subsample()
    assert scale > 0 and math.log2(scale) == round(math.log2(scale)) and clipped == 0
    flops = {}
    for policy in (17, 17 | 32):
        eng.set_fp8_policy(policy)
        eng.reset_timings()
        eng.encode(x, lens, case.beam)
        flops[policy] = (eng.timing("gemm")["flops"], eng.timing("gemm_fp8")["flops"])
    d = case.cfg["encoder_conf"]["output_size"]
    assert flops[17 | 32][1] - flops[17][1] == flops[17][0] - flops[17 | 32][0] > 2.0 * 9 * d * d * 100        # conv2's FLOPs changed sides
    m0 = _tap_metrics(eng, case, 0, 0)
    ref = RefBf16(name)
    fm = ref.frame_disagreement(eng)
    ter = _ter(eng.search(MODES, case.ctc_weight, case.reverse_weight), case)
    _record(case=name, dtype="fp8", policy="feed-forward + subsampling conv2", ter={m: list(v) for m, v in ter.items()}, chunk0=m0, frames=fm,
            reference_bf16_ter=ref.ter)
    assert eng.fp8_subsample()[1] == 0                               # the calibration batch itself never clips (2x headroom)
    assert m0["cos"] > 0.995, m0
    _assert_reduced_precision(name, "fp8 + conv2", ter, fm, ref, slack=FP8_SLACK)
    with pytest.raises(_lib.RvbError, match="6-bit"):
        eng.set_fp8_policy(64)
    eng.close()


def test_fp8_needs_dims_of_128():
    from golden_util import Case
    from reverb_amd._lib import RvbError
    from reverb_amd.engine import Engine
    case = Case("tiny_ln")                                  # d = 32
    with pytest.raises(RvbError, match="multiples of 128"):
        Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)


def test_fp8_bench_workload_against_reference():
    """bench.py --dtype fp8, checked: the hour of audio bench.py decodes, from PCM, against the reference's tokens -- the
    default policy (feed-forward GEMMs in fp8) within 1.5 x the reference's own bf16 behaviour; then every GEMM group in fp8
    (rvb_set_fp8_policy(31), round 2's mode): measured and recorded, no parity bound claimed."""
    from golden_util import LongCase
    from reverb_amd.engine import Engine
    from test_longform_gpu import MODES, RefBf16, _assert_reduced_precision, _record, _tap_metrics, _ter
    case = LongCase("r640_1h")
    n = len(case.js["lens"])
    ref = RefBf16("r640_1h")
    eng = Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.upload_pcm(case.pcm)
    nf = eng.fbank()
    eng.decode_resident(nf, ["ctc_greedy_search"], case.chunk, case.beam, case.ctc_weight, case.reverse_weight)     # calibration (bf16)
    got = {}
    for policy in ("feed-forward", "all groups"):
        if policy == "all groups":
            eng.set_fp8_policy(31)
        res = eng.decode_resident(nf, MODES, case.chunk, case.beam, case.ctc_weight, case.reverse_weight)
        ter = _ter(res, case)
        m_last = _tap_metrics(eng, case, n - 1, n - 1)
        fm = ref.frame_disagreement(eng)
        _record(case="r640_1h", dtype="fp8", policy=policy, chunks=n, ter={m: list(v) for m, v in ter.items()}, last_chunk=m_last,
                frames=fm, reference_bf16_ter=ref.ter)
        assert m_last["cos"] > 0.99, m_last
        got[policy] = (ter, fm)
    _assert_reduced_precision("r640_1h", "fp8 feed-forward", *got["feed-forward"], ref, slack=FP8_SLACK)
    # every group in fp8 does NOT meet that bar (measured: 16.2 % / 13.5 % token errors, 1 603 confident frames flipped against the
    # reference-bf16's 314) -- which is why it is not the default and why NO parity bound is claimed for it (round 4: the
    # literal 25 % that used to stand here was not a parity statement).  It is measured and recorded (profiles/*parity_metrics*),
    # its encoder output must stay an approximation of the reference's (cos > 0.99 above), and it must be worse than the default
    # on both measures -- otherwise the default should change
    ter_all, fm_all = got["all groups"]
    for m in MODES:
        assert ter_all[m][0] >= got["feed-forward"][0][m][0]
    assert fm_all["engine_confident"] >= got["feed-forward"][1]["engine_confident"]
    eng.close()


def test_fp8_policy_is_validated():
    from golden_util import LongCase
    from reverb_amd._lib import RvbError
    from reverb_amd.engine import Engine
    case = LongCase("r640_chunk")
    eng = Engine(case.cfg, case.sd, dtype="bf16", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    with pytest.raises(RvbError, match="RVB_FP8"):
        eng.set_fp8_policy(["ffn"])
    eng.close()


def test_fp8_scales_can_be_shared_between_engines():
    """rvb_get_fp8_scales / rvb_set_fp8_scales (the ranks of a sharded run must not calibrate independently): the
    element-wise maximum of what two engines calibrate on the two halves of a recording equals what one engine calibrates on
    the whole (max |.| over a union), and an engine that has the scales installed runs its FIRST batch in fp8."""
    from golden_util import LongCase
    from reverb_amd.engine import Engine
    case = LongCase("small_66")
    n = 8
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(n)])
    lens = np.array(case.js["lens"][:n], np.int32)
    mk = lambda: Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    whole, a, b = mk(), mk(), mk()
    assert whole.fp8_scales() is None
    whole.encode(x, lens, case.beam)
    a.encode(x[:4], lens[:4], case.beam)
    b.encode(x[4:], lens[4:], case.beam)
    sw, sa, sb = whole.fp8_scales(), a.fp8_scales(), b.fp8_scales()
    assert sw.shape == (case.cfg["encoder_conf"]["num_blocks"], 7) and (sw > 0).all()
    np.testing.assert_array_equal(np.maximum(sa, sb), sw)
    c = mk()
    c.set_fp8_scales(np.maximum(sa, sb))
    c.reset_timings(); c.set_profiling(True)
    c.encode(x, lens, case.beam)                         # no calibration pass: fp8 GEMMs on the very first batch
    assert c.timing("gemm_fp8")["launches"] > 0
    whole.encode(x, lens, case.beam)                     # second batch of `whole`: fp8 with the same scales
    got = [list(r.tokens) for r in c.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"]]
    want = [list(r.tokens) for r in whole.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"]]
    assert got == want
    for e in (whole, a, b, c):
        e.close()


def test_fp8_saturation_is_counted_not_silent():
    """VERDICT r3 "weak" #3: values beyond the calibrated range were clipped at +-448 silently.  rvb_get_fp8_saturation (round 4)
    counts what the kernels that write fp8 operands clip, per block and activation slot: zeros when an engine decodes the batch it
    calibrated on (2x headroom), and non-zero -- in the slots of the default policy's feed-forward GEMMs -- when scales a
    thousand times too small are installed; the counters reset on request and when scales are installed."""
    from golden_util import LongCase
    from reverb_amd.engine import Engine
    case = LongCase("small_66")
    n = 4
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(n)])
    lens = np.array(case.js["lens"][:n], np.int32)
    eng = Engine(case.cfg, case.sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(x, lens, case.beam)                       # calibration pass (bf16)
    eng.encode(x, lens, case.beam)                       # fp8 with the calibrated scales, same batch
    sat = eng.fp8_saturation()
    blocks = case.cfg["encoder_conf"]["num_blocks"]
    assert sat.shape == (blocks, 7) and sat.dtype == np.uint32
    assert int(sat.sum()) == 0, sat
    good = eng.fp8_scales()
    eng.set_fp8_scales(good / 1024.0)                    # everything above 448 / 1024 of the calibrated range now clips
    eng.encode(x, lens, case.beam)
    sat = eng.fp8_saturation(reset=True)
    # default policy = the two feed-forward modules: LayerNorm outputs (slots 0, 5) and the fp8 hidden activations (slots 1, 6)
    for slot in (0, 1, 5, 6):
        assert int(sat[:, slot].sum()) > 0, (slot, sat)
    for slot in (2, 3, 4):
        assert int(sat[:, slot].sum()) == 0, (slot, sat)  # qkv / pointwise convolutions stay bf16 under the default policy
    assert int(eng.fp8_saturation().sum()) == 0          # reset
    eng.set_fp8_scales(good)
    eng.encode(x, lens, case.beam)
    assert int(eng.fp8_saturation().sum()) == 0
    eng.close()


@pytest.mark.parametrize("B,Fi,Ti,Cin,Cout,stride,relu,use_res", [
    (2, 20, 30, 128, 128, 1, 1, True),        # the 128-channel stage: one K step per tap, two row tiles + a partial one
    (1, 10, 27, 256, 256, 1, 1, True),        # the 256-channel stage: two K steps per tap, 4 x 2 blocks per wave
    (2, 21, 31, 128, 256, 2, 1, False),       # the stride-2 convolution that opens stage 4
    (1, 9, 40, 128, 128, 1, 0, False),        # no ReLU (signed outputs: the amax is of |.|)
])
def test_fp8_implicit_gemm_convolution_against_fp64_on_the_quantised_values(lib, B, Fi, Ti, Cin, Cout, stride, relu, use_res):
    """conv_gemm.hip conv_igemm8_kernel (the MFMA-bound stages of the ResNet34 trunk on the fp8 path): a 3x3 /
    pad 1 convolution on e4m3 operands against fp64 on exactly the values those operands stand for -- gather, operand layout of
    the 32x32x64 scaled MFMA, block-to-slab mapping, scales, bias, bf16 residual, ReLU, both output forms, the recorded maximum."""
    rng = np.random.default_rng(B + Fi + Ti + Cin + Cout)
    x = f32(np.abs(rng.standard_normal((B, Fi, Ti, Cin))) * 1.3)               # post-ReLU activations
    w = f32(rng.standard_normal((Cout, 9, Cin)) / math.sqrt(9 * Cin) * (1 + rng.random((Cout, 1, 1)) * 3))
    bias = f32(rng.standard_normal(Cout) * 0.3)
    Fo, To = (Fi - 1) // stride + 1, (Ti - 1) // stride + 1
    res = f32(rng.standard_normal((B, Fo, To, Cout))) if use_res else None
    a_scale = float(2.0 ** math.ceil(math.log2(np.abs(x).max() * 2 / 448)))
    xd, wd = np.empty_like(x), np.empty_like(w)
    out = np.full((B, Fo, To, Cout), np.nan, np.float32)
    out8 = np.full((B, Fo, To, Cout), np.nan, np.float32)
    amax = np.zeros(1, np.float32)
    # reference on the de-quantised operands (filled by a first call), then the output scale from it
    out8_scale = 1.0
    for attempt in range(2):
        _lib.check(lib.rvb_test_conv_igemm_fp8(fptr(x), fptr(w), fptr(bias), fptr(res), fptr(out), fptr(out8), B, Fi, Ti, Cin, Cout, stride,
                                               relu, a_scale, out8_scale, fptr(xd), fptr(wd), fptr(amax)))
        xp = np.zeros((B, Fi + 2, Ti + 2, Cin), np.float64)
        xp[:, 1:-1, 1:-1] = xd
        v = np.zeros((B, Fo, To, Cout), np.float64)
        for kh in range(3):
            for kw in range(3):
                patch = xp[:, kh:kh + stride * (Fo - 1) + 1:stride, kw:kw + stride * (To - 1) + 1:stride]
                v += patch @ wd[:, kh * 3 + kw].astype(np.float64).T
        v += bias
        if use_res:
            from util import bf16_round
            v += bf16_round(res).astype(np.float64)
        if relu:
            v = np.maximum(v, 0)
        out8_scale = float(2.0 ** math.ceil(math.log2(max(np.abs(v).max(), 1e-6) * 2 / 448)))
    tol = 2e-4 * np.abs(v).max() + 1e-2 * np.abs(v)                            # scaled-MFMA K reduction + one bf16 rounding
    assert np.all(np.abs(out - v) <= tol), float(np.abs(out - v).max())
    grid = _e4m3_grid() * out8_scale
    near = grid[np.abs(grid[None, :] - v.reshape(-1, 1)[:4000]).argmin(1)]      # e4m3 output: the nearest representable value
    step = np.maximum(np.abs(near) / 8, out8_scale * 2.0 ** -9)
    assert np.all(np.abs(out8.reshape(-1)[:4000] - v.reshape(-1)[:4000]) <= step + tol.reshape(-1)[:4000])
    assert abs(float(amax[0]) - float(np.abs(out).max())) <= 1e-2 * float(np.abs(v).max())

