"""Parity of the HIP diarization networks (rvd_* C ABI) with the oracle's restatement of the
pyannote architectures (oracle/diar_ref.py; parity unpinned -- pyannote.audio is not available
here, see the oracle's header) on seeded synthetic weights."""
import os

import numpy as np
import pytest
import torch

from conftest import HAVE_GPU
from oracle import diar_ref as R
from reverb_amd import synth_diar as SD

pytestmark = pytest.mark.gpu


def _record(**kw):
    """measured parity numbers of this run -> gpurun_out/parity_metrics.jsonl (copied to profiles/ when judged)"""
    import json
    from conftest import ROOT
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def windows_of(pcm, cfg):
    wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
    n, win, step = wav.shape[0], cfg["window_samples"], cfg["step_samples"]
    full = (n - win) // step + 1 if n >= win else 0
    tail = n < win or (n - win) % step > 0
    W = full + (1 if tail else 0)
    x = torch.zeros(W, 1, win)
    for w in range(W):
        seg = wav[w * step:w * step + win]
        x[w, 0, :seg.shape[0]] = seg
    return x


@pytest.fixture(scope="module")
def case():
    cfg = SD.make_diar_config()
    seg_sd = SD.make_segmentation_sd(cfg, 0)
    pcm = SD.synth_conversation(14.3, seed=11)
    x = windows_of(pcm, cfg)
    taps = {}
    with torch.no_grad():
        logp = R.pyannet(R.to_torch_sd(seg_sd), x, taps)
    return dict(cfg=cfg, seg_sd=seg_sd, pcm=pcm, x=x, logp=logp.numpy(), taps={k: v.numpy() for k, v in taps.items()})


def test_window_count_matches_pyannote_slide(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    assert eng.frames == R.NUM_FRAMES
    for n, want in ((1, 1), (159999, 1), (160000, 1), (160001, 2), (176000, 2), (176001, 3), (228800, 6), (57600000, 3591)):
        assert eng.num_windows(n) == want, n
    eng.close()


def test_segmentation_f32_matches_oracle(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    W = eng.upload(case["pcm"])
    assert W == case["x"].shape[0] == 6
    logp = eng.segment()
    sinc = eng.tap("sincnet", W)
    lstm = eng.tap("lstm", W)
    # fp32 everywhere; differences are summation order (sinc bank shared across windows, MFMA K order)
    assert np.abs(sinc - case["taps"]["sincnet"]).max() < 2e-3
    assert np.abs(lstm - case["taps"]["lstm"]).max() < 2e-3
    assert np.abs(logp - case["logp"]).max() < 1e-2
    agree = (logp.argmax(-1) == case["logp"].argmax(-1)).mean()
    assert agree > 0.995, agree
    assert np.allclose(np.exp(logp).sum(-1), 1.0, atol=1e-4)
    eng.close()


def test_sinc_filter_bank_on_the_matrix_pipe_equals_the_vector_form(case, monkeypatch, lab):
    """diar.hip sinc_mfma_kernel (round 5, default: frames x taps x filters on v_mfma_f32_32x32x2_f32, exact fp32, taps paired
    (j, j + 126)) against the VALU form (lab switch RVD_SINC_MFMA=0, taps in order): fp32 both, only the summation order
    differs -- the SincNet output (three layers further down) agrees to 1e-4 of its scale."""
    from reverb_amd.diar_engine import DiarEngine
    taps = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVD_SINC_MFMA", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
        W = eng.upload(case["pcm"])
        eng.segment()
        taps[flag] = eng.tap("sincnet", W)
        eng.close()
    scale = np.abs(taps["0"]).max()
    assert np.abs(taps["0"] - taps["1"]).max() < 1e-4 * scale, np.abs(taps["0"] - taps["1"]).max() / scale
    assert np.abs(taps["1"] - case["taps"]["sincnet"]).max() < 2e-3


def test_sincnet_thin_gemm_and_vector_pool_norm_equal_the_forms_they_replaced(case, monkeypatch, lab):
    """Round 6, bf16: SincNet's conv layers 2 / 3 on conv1d5 (weights resident in LDS, 256-frame tiles) against the generic GEMM
    (RVD_CONV1D5=0) and pool_norm in 16-byte vectors against the scalar form (RVD_POOLNORM_VEC=0).  The same bf16 products summed in
    another order (and fp64 statistics added in another order): the SincNet output, three bf16 roundings further down, agrees to
    bf16 resolution, and both stay as close to the fp32 oracle as the bf16 segmentation test asks."""
    from reverb_amd.diar_engine import DiarEngine
    taps = {}
    for conv, vec in (("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")):
        monkeypatch.setenv("RVD_CONV1D5", conv)
        monkeypatch.setenv("RVD_POOLNORM_VEC", vec)
        eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
        W = eng.upload(case["pcm"])
        eng.segment()
        taps[conv + vec] = eng.tap("sincnet", W)
        eng.close()
    ref = taps["00"]
    scale = np.abs(ref).max()
    for k in ("10", "01", "11"):
        d = np.abs(taps[k] - ref)
        assert d.max() < 0.03 * scale and d.mean() < 2e-3 * scale, (k, d.max() / scale, d.mean() / scale)
    o = case["taps"]["sincnet"]
    e_old, e_new = np.abs(ref - o).mean(), np.abs(taps["11"] - o).mean()
    assert e_new < 1.1 * e_old + 1e-6, (e_old, e_new)


def test_segmentation_f32_batching_is_invariant(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    eng.upload(case["pcm"])
    a = eng.segment(batch=512)
    b = eng.segment(batch=4)            # 4 + 2 windows
    c = eng.segment(first=3, n=2)
    assert np.array_equal(a, b)
    assert np.array_equal(a[3:5], c)
    eng.close()


def test_rerun_resident_reproduces_the_upload(case):
    """rvd_rerun_resident (the timed step of bench_diar.py starts from HBM-resident samples): the front end run again on the
    recording that is already on the device gives the same windows and bit-identical segmentation output; before any upload
    it is refused."""
    from reverb_amd.diar_engine import DiarEngine
    from reverb_amd._lib import RvbError
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    with pytest.raises(RvbError):
        eng.rerun_resident()
    W = eng.upload(case["pcm"])
    a = eng.segment()
    assert eng.rerun_resident() == W
    b = eng.segment()
    assert np.array_equal(a, b)
    eng.close()


def test_segmentation_bf16_close_to_oracle(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
    W = eng.upload(case["pcm"])
    logp = eng.segment()
    sinc = eng.tap("sincnet", W)
    ref = case["taps"]["sincnet"]
    assert np.abs(sinc - ref).mean() < 0.02 * np.abs(ref).mean() + 1e-3
    # bf16 inputs through 4 recurrent layers of 589 steps: compare distributions, not bits
    p, q = np.exp(logp), np.exp(case["logp"])
    assert np.abs(p - q).mean() < 0.02
    agree = (logp.argmax(-1) == case["logp"].argmax(-1)).mean()
    assert agree > 0.93, agree
    eng.close()


def test_short_audio_single_padded_window(case):
    from reverb_amd.diar_engine import DiarEngine
    pcm = case["pcm"][:52345]
    x = windows_of(pcm, case["cfg"])
    with torch.no_grad():
        want = R.pyannet(R.to_torch_sd(case["seg_sd"]), x).numpy()
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    assert eng.upload(pcm) == 1
    got = eng.segment()
    assert np.abs(got - want).max() < 1e-2
    eng.close()


# ------------------------------------------------------------------------------------ embedding network
@pytest.fixture(scope="module")
def emb_case(case):
    cfg = case["cfg"]
    emb_sd = SD.make_embedding_sd(cfg, 0)
    rng = np.random.default_rng(5)
    W = case["x"].shape[0]
    wins, masks = [], []
    for w in (0, 2, W - 1):                       # W-1 is the zero-padded tail window
        for k in range(3):
            m = np.zeros(R.NUM_FRAMES, np.float32)
            if k == 0:
                m[rng.integers(0, 200):rng.integers(300, 589)] = 1.0        # one contiguous turn
            elif k == 1:
                m[:] = (rng.random(R.NUM_FRAMES) < 0.4)                       # scattered frames
            elif w == 2:
                pass                                                          # inactive speaker: all-zero mask
            else:
                m[:] = rng.random(R.NUM_FRAMES).astype(np.float32)           # soft weights
            wins.append(w); masks.append(m)
    wins = np.array(wins, np.int64); masks = np.stack(masks)
    sd = R.to_torch_sd(emb_sd)
    feats = {w: torch.from_numpy(R.hamming_fbank(case["x"][w, 0].numpy())) for w in sorted(set(wins.tolist()))}
    with torch.no_grad():
        trunk = {w: R.resnet34_trunk(sd, f[None]) for w, f in feats.items()}
        want = torch.cat([F_linear_stats(sd, trunk[int(w)], torch.from_numpy(m)[None]) for w, m in zip(wins, masks)]).numpy()
    return dict(cfg=cfg, emb_sd=emb_sd, wins=wins, masks=masks, feats={w: f.numpy() for w, f in feats.items()}, want=want)


def F_linear_stats(sd, trunk, weights):
    return torch.nn.functional.linear(R.tstp(trunk, weights), sd["resnet.seg_1.weight"], sd["resnet.seg_1.bias"])


def test_embedding_fbank_shared_across_windows(case, emb_case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="f32")
    eng.upload(case["pcm"])
    for w, want in emb_case["feats"].items():
        got = eng.emb_fbank(w)
        assert got.shape == (998, 80)
        got = got - got.mean(0, keepdims=True)        # the oracle returns mean-normalised features
        assert np.abs(got - want).max() < 2e-3, w
    eng.close()


def test_embedding_f32_matches_oracle(case, emb_case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="f32")
    eng.upload(case["pcm"])
    got = eng.embed(emb_case["wins"], emb_case["masks"])
    want = emb_case["want"]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 2e-3 * scale, np.abs(got - want).max() / scale
    # an inactive speaker (all-zero mask) pools to zero statistics: the embedding is the bias of seg_1
    idx = int(np.where(emb_case["masks"].sum(1) == 0)[0][0])
    assert np.abs(got[idx] - emb_case["emb_sd"]["resnet.seg_1.bias"]).max() < 1e-5
    # item order / grouping does not matter
    perm = np.array([3, 4, 5, 0, 1, 2, 6, 7, 8])
    again = eng.embed(emb_case["wins"][perm], emb_case["masks"][perm])
    assert np.array_equal(again, got[perm])
    eng.close()


def test_embedding_bf16_close_to_oracle(case, emb_case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
    eng.upload(case["pcm"])
    got = eng.embed(emb_case["wins"], emb_case["masks"])
    want = emb_case["want"]
    active = emb_case["masks"].sum(1) > 0
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos[active].min() > 0.999, cos          # SURVEY 8(d)'s bar for embeddings; measured 0.99999
    eng.close()


def test_embedding_retries_with_fewer_windows_when_the_workspace_does_not_fit(case, emb_case, monkeypatch, lab):
    """the E_NOMEM retry of embed_impl (halve the windows per pass, release, regroup) was dead code -- HIP's sticky
    last error failed the retried pass, and a request below the configured batch never shrank.  The lab hook
    RVD_FAKE_NOMEM_ABOVE makes every workspace request of more than one window fail: the three windows of the case must
    come out of three one-window passes, bit-identical to the one-pass run."""
    from reverb_amd.diar_engine import DiarEngine
    out = {}
    for fake in (None, "1"):
        if fake:
            monkeypatch.setenv("RVD_FAKE_NOMEM_ABOVE", fake)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        out[fake] = eng.embed(emb_case["wins"], emb_case["masks"])
        assert (eng.emb_windows_per_pass() == 1) == bool(fake)
        eng.close()
    assert np.array_equal(out[None], out["1"])


def test_implicit_gemm_convolutions_equal_the_direct_kernel(case, emb_case, monkeypatch, lab):
    """conv_gemm.hip (stages 3-4 of the ResNet34, default) against resnet.hip's direct convolution kernel on the same
    bf16 weights and inputs: the two differ only in fp32 summation order, so the embeddings agree far tighter than
    either does with the fp32 oracle; the timing keys prove that both kernels really ran."""
    from reverb_amd.diar_engine import DiarEngine
    out, flops, ig = {}, {}, {}
    monkeypatch.setenv("RVD_CONV_SC_FUSE", "0")      # the shortcut as a kernel of its own in all three runs: the FLOP counts compare
    for flag in ("0", "1", "2"):            # 2 = the default since round 4: the stride-2 convolutions of stages 3-4 as well
        monkeypatch.setenv("RVD_CONV_IGEMM", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        flops[flag] = sum(eng.timing(k)[1] for k in ("emb_conv_128", "emb_conv_256", "emb_conv_s2_128", "emb_conv_s2_256"))
        ig[flag] = eng.timing("emb_conv_igemm")[2]
        eng.close()
    assert flops["0"] == flops["1"] == flops["2"] > 0
    assert ig["0"] == 0 and ig["1"] >= 16          # 11 + 5 stride-1 3x3 convolutions of stages 3-4 per trunk pass
    assert ig["2"] == ig["1"] // 16 * 18           # + the two stride-2 convolutions that open those stages
    active = emb_case["masks"].sum(1) > 0
    want = emb_case["want"][active]
    for flag in ("1", "2"):
        a, b = out["0"][active], out[flag][active]
        cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        assert cos.min() > 0.9995, (flag, cos)
        cosw = (b * want).sum(1) / (np.linalg.norm(b, axis=1) * np.linalg.norm(want, axis=1))
        assert cosw.min() > 0.995, (flag, cosw)


def test_wide_implicit_gemm_tile_equals_the_narrow_one(case, emb_case, monkeypatch, lab):
    """The 128-channel stage on 512-pixel tiles (wave tile 128 x 64, default for launches of >= 512 Ki pixels) against the
    256-pixel tile: the K order of every output is the same, so the embeddings are IDENTICAL; the counter proves which ran."""
    from reverb_amd.diar_engine import DiarEngine
    out, wide = {}, {}
    for bm in ("256", "512"):
        monkeypatch.setenv("RVD_IGEMM_BM", bm)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[bm] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        wide[bm] = eng.timing("emb_conv_igemm_wide")[2]
        eng.close()
    assert wide["256"] == 0 and wide["512"] >= 10, wide      # 11 stride-1 convolutions of stage 3 (one carries the shortcut) + the stride-2 one
    assert np.array_equal(out["256"], out["512"])


def test_streamed_convolutions_equal_the_direct_kernel(case, emb_case, monkeypatch, lab):
    """conv_stream.hip (the stride-1 convolutions of the 32- and 64-channel stages as a stream of tiles per workgroup: weights
    resident in LDS, patches by LDS-DMA ahead of the MFMAs, counted vmcnt) against resnet.hip's one-tile-per-workgroup kernel:
    same operand values, accumulation order and rounding points -- the embeddings must be IDENTICAL, whatever the split of
    the time axis (1, 2, 3 or 17 workgroups per row of tiles: 17 = one tile each, the pipeline never reaches its steady
    state), and the counters prove which path ran.  The windows include the zero-padded tail window; 998 frames = 16 full
    tiles of 62 + one of 6, 80 mel rows = 20 row tiles, the last patch rows and columns clamped to the bordered plane."""
    from reverb_amd.diar_engine import DiarEngine
    out, streamed, flops = {}, {}, {}
    monkeypatch.setenv("RVD_CONV_STREAM64", "1")            # the 64-channel stage too (opt-in: measured equal to the direct kernel)
    monkeypatch.setenv("RVD_CONV_BLOCK", "0")                # stage 1 as single convolutions (the default fuses its blocks: conv_block.hip)
    monkeypatch.setenv("RVD_CONV_ROW64", "0")                # stage 2 on the kernel under test, not on conv_row64.hip
    for flag in ("0", "1", "2", "3", "17"):
        monkeypatch.setenv("RVD_CONV_STREAM", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        streamed[flag] = eng.timing("emb_conv_stream")[2]
        flops[flag] = eng.timing("emb_conv_32")[1] + eng.timing("emb_conv_64")[1]
        eng.close()
    assert streamed["0"] == 0
    for flag in ("1", "2", "3", "17"):
        assert streamed[flag] >= 6 + 7, streamed        # 6 convolutions of stage 1 + 7 stride-1 ones of stage 2, per trunk pass
        assert flops[flag] == flops["0"] > 0
        assert np.array_equal(out["0"], out[flag]), flag


def test_projection_shortcut_fused_into_the_second_convolution(case, emb_case, monkeypatch, lab):
    """Default since round 5 (first run on a GPU there; lab switch RVD_CONV_SC_FUSE=0 = the separate kernel): the 1x1 / stride-2 projection shortcut of the blocks that open stages 3 and 4 becomes one
    or two extra K steps of the block's second convolution (conv_gemm.hip, ConvArgs::in2) -- no shortcut kernel, no residual tensor.
    Not bit-identical by construction (the shortcut's output is no longer rounded to bf16 before it is added), so: the embeddings
    agree with the unfused run far tighter than either does with the fp32 oracle, and the counters prove which path ran."""
    from reverb_amd.diar_engine import DiarEngine
    out, fused, sc = {}, {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVD_CONV_SC_FUSE", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        fused[flag] = eng.timing("emb_conv_sc_fused")[2]
        sc[flag] = eng.timing("emb_conv_sc")[2]
        eng.close()
    assert fused["0"] == 0 and fused["1"] >= 2                # stages 3 and 4, per trunk pass
    assert sc["1"] == sc["0"] - fused["1"] == 0               # (stage 2's shortcut is written by conv_s2.hip beside the stride-2 convolution)
    active = emb_case["masks"].sum(1) > 0
    a, b, want = out["0"][active], out["1"][active], emb_case["want"][active]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert cos.min() > 0.9995, cos
    cosw = (b * want).sum(1) / (np.linalg.norm(b, axis=1) * np.linalg.norm(want, axis=1))
    assert cosw.min() > 0.995, cosw


def test_trunk_stages_3_and_4_in_fp8(case, emb_case):
    """DiarEngine(dtype="fp8") = rvd_model_cfg.dtype RVB_FP8 (BASELINE configs[4]; first run on a GPU in round 5): stages 3-4 of the ResNet34 trunk on e4m3 operands (conv_igemm8_kernel), per-tensor
    activation scales from the first trunk pass.  The first embed() call calibrates (bf16: it must equal the bf16 engine's result
    exactly); the second runs the fp8 kernels (counter), clips nothing of what it was calibrated on, and its embeddings stay close
    to the bf16 ones and to the fp32 oracle."""
    from reverb_amd.diar_engine import DiarEngine
    ref = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
    ref.upload(case["pcm"])
    want16 = ref.embed(emb_case["wins"], emb_case["masks"])
    assert ref.emb_fp8()[0] == 0
    ref.close()
    eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="fp8")
    eng.upload(case["pcm"])
    assert eng.emb_fp8()[0] == 0
    first = eng.embed(emb_case["wins"], emb_case["masks"])                    # calibration pass
    assert np.array_equal(first, want16)
    state, scales, clipped = eng.emb_fp8()
    assert state == 2 and clipped == 0
    used = scales[[((li - 2) * 8 + bi) * 2 + k for li, nb in ((2, 6), (3, 3)) for bi in range(nb) for k in (0, 1)]]
    assert np.all(used > 0) and np.all(np.log2(used) == np.round(np.log2(used)))
    eng.reset_timings()
    got = eng.embed(emb_case["wins"], emb_case["masks"])
    assert eng.timing("emb_conv_fp8")[2] == 2 * (6 + 3) - 2                   # every 3x3 convolution of stages 3-4 but the two stride-2 ones
    assert eng.emb_fp8()[2] == 0
    active = emb_case["masks"].sum(1) > 0
    a, b, want = want16[active], got[active], emb_case["want"][active]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    cosw = (b * want).sum(1) / (np.linalg.norm(b, axis=1) * np.linalg.norm(want, axis=1))
    cosw16 = (a * want).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(want, axis=1))
    _record(test="trunk_stages_3_and_4_in_fp8", cos_fp8_vs_bf16_min=float(cos.min()), cos_fp8_vs_bf16_mean=float(cos.mean()),
            cos_fp8_vs_fp32_oracle_min=float(cosw.min()), cos_bf16_vs_fp32_oracle_min=float(cosw16.min()), embeddings=int(active.sum()))
    assert cos.min() > 0.999, cos              # SURVEY 8(d)'s bar; measured 0.99960 (profiles/archive/r05h_parity_metrics.jsonl)
    assert cosw.min() > 0.999, cosw            # measured 0.99960; the bf16 engine: 0.99999
    eng.close()



def test_row64_convolutions_equal_the_direct_kernel(case, emb_case, monkeypatch, lab):
    """conv_row64.hip (round 5, default for the stride-1 3x3 convolutions of the 64-channel stage: weights in registers, a lane's
    accumulators = 8 consecutive channels of one pixel, swizzled patch, two workgroups per CU) against resnet.hip's direct kernel
    (lab switch RVD_CONV_ROW64=0): same operand values, accumulation order (chunks outer, taps inner) and rounding points -- the
    embeddings must be IDENTICAL; the counters prove which path ran.  The windows include the zero-padded tail window; 499 frames
    = 16 full tiles of 30 + one of 19, 40 mel rows = 10 row tiles, with and without a residual."""
    from reverb_amd.diar_engine import DiarEngine
    out, row64, flops = {}, {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVD_CONV_ROW64", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        row64[flag] = eng.timing("emb_conv_row64")[2]
        flops[flag] = eng.timing("emb_conv_64")[1]
        eng.close()
    assert row64["0"] == 0 and row64["1"] >= 7                     # the 7 stride-1 convolutions of stage 2, per trunk pass
    assert flops["1"] == flops["0"] > 0
    assert np.array_equal(out["0"], out["1"])


def test_stride2_opener_equals_the_two_launches(case, emb_case, monkeypatch, lab):
    """conv_s2.hip (round 5, default): the stride-2 3x3 convolution (32 -> 64 channels) and the 1x1 / stride-2 projection shortcut
    of the block that opens the 64-channel stage in ONE pass over the block's input (de-interleaved patch planes, both outputs
    written by the same kernel) against two launches of resnet.hip's direct kernel (lab switch RVD_CONV_S2SC=0): same operand
    values, accumulation order (taps 0 .. 8) and rounding points -- the embeddings must be IDENTICAL; the counters prove which
    path ran and that the FLOPs credited are the same (stride-2 convolution + shortcut).  998 -> 499 frames = 16 tiles of 31 +
    one of 3, 80 -> 40 mel rows; the windows include the zero-padded tail window."""
    from reverb_amd.diar_engine import DiarEngine
    out, fused, flops = {}, {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVD_CONV_S2SC", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        fused[flag] = eng.timing("emb_conv_s2sc")[2]
        flops[flag] = eng.timing("emb_conv_s2_64")[1] + eng.timing("emb_conv_sc")[1]
        eng.close()
    assert fused["0"] == 0 and fused["1"] >= 1
    assert flops["1"] == flops["0"] > 0
    assert np.array_equal(out["0"], out["1"])


def test_fused_basic_block_equals_two_convolutions(case, emb_case, monkeypatch, lab):
    """conv_block.hip conv_block32_kernel (round 5, default): a whole stride-1 BasicBlock of the 32-channel stage per launch -- the
    intermediate tensor stays in LDS, the residual comes out of the input patch -- against the same block as two convolution
    launches (lab switch RVD_CONV_BLOCK=0): same operand values, same accumulation order, same rounding points, so the
    embeddings must be IDENTICAL; the counters prove which path ran and that the FLOPs credited are the same."""
    from reverb_amd.diar_engine import DiarEngine
    out, fused, flops, streamed = {}, {}, {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVD_CONV_BLOCK", flag)
        eng = DiarEngine(case["cfg"], case["seg_sd"], emb_case["emb_sd"], dtype="bf16")
        eng.upload(case["pcm"])
        eng.reset_timings(); eng.set_profiling(True)
        out[flag] = eng.embed(emb_case["wins"], emb_case["masks"])
        eng.set_profiling(False)
        fused[flag] = eng.timing("emb_conv_block")[2]
        flops[flag] = eng.timing("emb_conv_32")[1]
        streamed[flag] = eng.timing("emb_conv_stream")[2]
        eng.close()
    assert fused["0"] == 0 and fused["1"] >= 3                     # the three blocks of stage 1, per trunk pass
    assert streamed["0"] - streamed["1"] == 2 * fused["1"]         # each replaces two streamed convolutions
    assert flops["1"] == flops["0"] > 0
    assert np.array_equal(out["0"], out["1"])


def test_fused_basic_block_on_ragged_shapes(lib):
    """The fused block through librvb_test.so's hook on planes whose height is not a multiple of 3 (the rows of a step) and whose
    width is not a multiple of 60 (a partial last band, a single band, fewer rows than a step), against fp64 on the bf16-rounded operands with
    the intermediate rounded to bf16 as the kernel (and the unfused path) rounds it."""
    from reverb_amd import _lib
    from util import bf16_round, f32
    rng = np.random.default_rng(5)
    for B, F, T in ((2, 6, 61), (1, 9, 130), (1, 3, 17), (1, 4, 60), (1, 1, 5), (1, 2, 64), (1, 7, 121)):
        x = bf16_round(f32(np.abs(rng.standard_normal((B, F, T, 32)))))
        wa = bf16_round(f32(rng.standard_normal((32, 32, 3, 3)) / 12.0))
        wb = bf16_round(f32(rng.standard_normal((32, 32, 3, 3)) / 12.0))
        ba, bb = f32(rng.standard_normal(32) * 0.1), f32(rng.standard_normal(32) * 0.1)
        got = np.zeros((B, F, T, 32), np.float32)
        _lib.check(lib.rvb_test_conv_block32(_lib.fptr(x), _lib.fptr(wa), _lib.fptr(ba), _lib.fptr(wb), _lib.fptr(bb), _lib.fptr(got), B, F, T))

        def conv(inp, w, b):            # NHWC 3x3 pad 1, fp64
            p = np.pad(inp.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
            o = np.zeros(inp.shape[:3] + (32,))
            for kh in range(3):
                for kw in range(3):
                    o += p[:, kh:kh + F, kw:kw + T, :] @ w[:, :, kh, kw].astype(np.float64).T
            return o + b
        mid = bf16_round(f32(np.maximum(conv(x, wa, ba), 0.0)))
        want = np.maximum(conv(mid, wb, bb) + x, 0.0)
        err = np.abs(got - want)
        assert err.max() < 2e-2 * max(1.0, np.abs(want).max()), (B, F, T, err.max())
        assert np.mean(err > 1e-2 * np.abs(want).max()) < 1e-3, (B, F, T)      # bf16 output rounding (and rare double roundings of mid)


# ------------------------------------------------------------------------------------ clustering on the GPU
def test_stride2_opener_on_ragged_shapes(lib):
    """conv_s2.hip through librvb_test.so's hook on planes of odd and even height / width (partial row tiles, a partial last time
    tile, a single tile, one row), against fp64 on the bf16-rounded operands; the hook also checks the zero borders."""
    from reverb_amd import _lib
    from util import bf16_round, f32
    rng = np.random.default_rng(6)
    for B, Fi, Ti in ((2, 8, 62), (1, 9, 125), (1, 3, 17), (1, 1, 2), (1, 16, 63), (1, 5, 130)):
        Fo, To = (Fi - 1) // 2 + 1, (Ti - 1) // 2 + 1
        x = bf16_round(f32(rng.standard_normal((B, Fi, Ti, 32))))
        w = bf16_round(f32(rng.standard_normal((64, 32, 3, 3)) / 12.0))
        wsc = bf16_round(f32(rng.standard_normal((64, 32)) / 4.0))
        b, bsc = f32(rng.standard_normal(64) * 0.1), f32(rng.standard_normal(64) * 0.1)
        out, sc = np.zeros((B, Fo, To, 64), np.float32), np.zeros((B, Fo, To, 64), np.float32)
        _lib.check(lib.rvb_test_conv_s2sc(_lib.fptr(x), _lib.fptr(w), _lib.fptr(b), _lib.fptr(wsc), _lib.fptr(bsc), _lib.fptr(out), _lib.fptr(sc), B, Fi, Ti))
        xp = np.zeros((B, Fi + 2, Ti + 2, 32))
        xp[:, 1:-1, 1:-1] = x
        want = np.zeros((B, Fo, To, 64))
        for kh in range(3):
            for kw in range(3):
                want += np.einsum("bftc,oc->bfto", xp[:, kh:kh + 2 * Fo:2, kw:kw + 2 * To:2][:, :Fo, :To], w[:, :, kh, kw].astype(np.float64))
        want = np.maximum(want + b, 0.0)
        wsc_ = np.einsum("bftc,oc->bfto", x[:, ::2, ::2].astype(np.float64), wsc.astype(np.float64)) + bsc
        for got, ref in ((out, want), (sc, wsc_)):
            err = np.abs(got - ref)
            assert err.max() < 1e-2 * max(1.0, np.abs(ref).max()), (B, Fi, Ti, err.max())


@pytest.mark.parametrize("n,d,seed", [(2, 8, 0), (3, 4, 1), (257, 16, 2), (1500, 256, 3), (3100, 64, 4)])
def test_centroid_linkage_matches_scipy(case, n, d, seed):
    """scipy is what pyannote itself calls; it is installed on the GPU box, so the kernel is pinned to it.  n > 1024: several
    batches of slots per thread, re-dealt from the live list every 256 merges (the batch count drops on the way)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from reverb_amd.diar_engine import DiarEngine
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((5, d))
    X = centers[rng.integers(5, size=n)] + 0.35 * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    want = linkage(X, method="centroid", metric="euclidean")
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
    got = eng.centroid_linkage(X)
    eng.close()
    assert got.shape == want.shape
    assert np.array_equal(got[:, [0, 1, 3]], want[:, [0, 1, 3]])          # same merges in the same order
    assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9
    assert np.array_equal(fcluster(got, 0.7045654963945799, "distance"), fcluster(want, 0.7045654963945799, "distance"))


@pytest.mark.parametrize("n,d,seed", [(2, 8, 0), (3, 4, 1), (17, 8, 5), (257, 16, 2), (1500, 256, 3), (3100, 64, 4), (6000, 32, 6)])
def test_centroid_linkage_on_sixteen_workgroups_matches_scipy_and_the_one_workgroup_loop(case, monkeypatch, lab, n, d, seed):
    """Round 4: the merge loop on 16 persistent workgroups of one XCD (linkage.hip linkage_mb_kernel: one grid barrier per
    merge, exact nearest-neighbour candidates, agent-scope relaxed atomics through the XCD's L2).  RVD_LINKAGE_MB=1 forces it
    for any n (the default takes it from 3 000 points on).  Pinned to scipy like the one-workgroup loop -- same merges in
    the same order, distances < 1e-9 -- and BIT-IDENTICAL to the one-workgroup loop (RVD_LINKAGE_MB=0): both round the
    Lance-Williams update through the same function.  The status word tells that it really ran (no silent fall-back)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from reverb_amd.diar_engine import DiarEngine
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((5, d))
    X = centers[rng.integers(5, size=n)] + 0.35 * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    want = linkage(X, method="centroid", metric="euclidean")
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RVD_LINKAGE_MB", mode)
        monkeypatch.setenv("RVD_LINKAGE_PROF", "1")          # prints when the multi-workgroup loop gives up (stderr)
        eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
        got[mode] = eng.centroid_linkage(X)
        eng.close()
    assert np.array_equal(got["1"][:, [0, 1, 3]], want[:, [0, 1, 3]])
    assert np.abs(got["1"][:, 2] - want[:, 2]).max() < 1e-9
    assert np.array_equal(got["1"], got["0"])
    assert np.array_equal(fcluster(got["1"], 0.7045654963945799, "distance"), fcluster(want, 0.7045654963945799, "distance"))


def test_centroid_linkage_without_slot_compaction_matches_scipy(case, monkeypatch, lab):
    """10 240 < n <= ~11 500 points keep the LDS-resident state but not the re-dealt slot lists (registers for ten batches of
    1024): RVD_LINKAGE_COMPACT=0 runs that variant on a small input."""
    from scipy.cluster.hierarchy import linkage
    from reverb_amd.diar_engine import DiarEngine
    monkeypatch.setenv("RVD_LINKAGE_COMPACT", "0")
    rng = np.random.default_rng(21)
    X = rng.standard_normal((4, 48))[rng.integers(4, size=1300)] + 0.3 * rng.standard_normal((1300, 48))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
    got = eng.centroid_linkage(X)
    eng.close()
    want = linkage(X, method="centroid", metric="euclidean")
    assert np.array_equal(got[:, [0, 1, 3]], want[:, [0, 1, 3]])
    assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9


def test_centroid_linkage_large_n_variant_matches_scipy(case, monkeypatch, lab):
    """More than ~11 500 points (3 h of audio) do not fit the LDS-resident state: the same loop then keeps its state in
    global memory.  RVD_LINKAGE_GLOBAL forces that variant on a small input so that it is checked against scipy too."""
    from scipy.cluster.hierarchy import linkage
    from reverb_amd.diar_engine import DiarEngine
    monkeypatch.setenv("RVD_LINKAGE_GLOBAL", "1")
    rng = np.random.default_rng(12)
    X = rng.standard_normal((3, 32))[rng.integers(3, size=700)] + 0.3 * rng.standard_normal((700, 32))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
    got = eng.centroid_linkage(X)
    eng.close()
    want = linkage(X, method="centroid", metric="euclidean")
    assert np.array_equal(got[:, [0, 1, 3]], want[:, [0, 1, 3]])
    assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9
