"""Long-form parity on the GPU, against goldens of the unmodified reference decoded chunk by chunk (batch 1, as its CLI
does): `small_66` (66 chunks of a d=128 model: the B >= 64 two-slice pipeline), `r640_chunk` and `r640_1h` -- the exact
workload bench.py times (r640 weights with the frozen CTC calibration, 1 h of synth_audio(1234), 176 chunks, from PCM
through the device fbank, encoded as 144 + 32 chunk slices).

f32 mode must give the reference's token ids; bf16 mode is judged by token error rate against the reference, encoder
cosine similarity and CTC log-prob differences -- the bounds written below are the measured values of round 2 plus a
margin (profiles/r02_parity_metrics.jsonl holds the measurements)."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from golden_util import GOLDEN, LongCase
from reverb_amd.engine import Engine

pytestmark = pytest.mark.gpu
MODES = ["ctc_greedy_search", "attention_rescoring"]


def _edit_distance(a, b):
    a, b = list(a), list(b)
    dp = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, dp[0] = dp[0], i
        for j in range(1, len(b) + 1):
            cur = dp[j]
            dp[j] = min(dp[j] + 1, dp[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev = cur
    return dp[-1]


def _record(**kw):
    """Measured parity numbers of this run -> gpurun_out/parity_metrics.jsonl (copied to profiles/ when judged)."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _ter(results, case):
    out = {}
    for m in MODES:
        err = tot = bad_chunks = 0
        for r, g in zip(results[m], case.golden(m)):
            e = _edit_distance(r.tokens, g["tokens"])
            err += e
            tot += len(g["tokens"])
            bad_chunks += e > 0
        out[m] = (err, tot, bad_chunks)
    return out


def _tap_metrics(eng, case, first_chunk_index_in_batch, c):
    """encoder cos-sim / max abs and top-1 CTC log-prob differences of chunk `c` (engine batch row given) vs the reference."""
    n = case.js["encoder_lens"][c]
    enc = eng.encoder_out()[first_chunk_index_in_batch, :n:16, ::8].astype(np.float64)
    gold = case.arrays[f"encoder_out_{c}"].astype(np.float64)
    cos = float((enc * gold).sum() / (np.linalg.norm(enc) * np.linalg.norm(gold)))
    tv, ti = eng.ctc_topk()
    v, i = tv[first_chunk_index_in_batch, :n, 0], ti[first_chunk_index_in_batch, :n, 0]
    same = i == case.arrays[f"ctc_argmax_{c}"]
    d = np.abs(v - case.arrays[f"ctc_top1_{c}"])[same]
    return dict(cos=cos, enc_max_abs=float(np.abs(enc - gold).max()), argmax_agree=float(same.mean()),
                logp_mean_abs=float(d.mean()), logp_p99_abs=float(np.quantile(d, 0.99)), logp_max_abs=float(d.max()))


class RefBf16:
    """What bf16 costs the REFERENCE ITSELF on this case (oracle/gen_golden_bf16ref.py: the unmodified reference under
    torch.autocast('cpu', bfloat16) against its own fp32 run): the yardstick of the engine's reduced-precision modes."""

    def __init__(self, name):
        with open(os.path.join(GOLDEN, name + "_refbf16.json")) as f:
            self.js = json.load(f)
        self.arrays = np.load(os.path.join(GOLDEN, name + "_refbf16.npz"))
        self.ter = self.js["ter"]
        self.tokens = self.js["ref_tokens"]

    def ter_bound(self, mode):
        """reference-bf16 TER + a margin for the different summation order (1 % of the tokens, at least 3 tokens)."""
        return self.ter[mode] + max(0.01, 3.0 / self.tokens[mode])

    def frame_disagreement(self, eng, first_chunk=0):
        """Frames whose argmax differs from the fp32 reference's, over the engine's current batch: (all frames, frames where
        the reference's own top-2 log-prob margin exceeds 0.1) for the engine and for the reference under bf16 autocast."""
        _, ti = eng.ctc_topk()
        nb = ti.shape[0]
        a32 = self.arrays["argmax_f32"][first_chunk:first_chunk + nb].astype(np.int64)
        abf = self.arrays["argmax_bf16"][first_chunk:first_chunk + nb].astype(np.int64)
        gap = self.arrays["gap_f32"][first_chunk:first_chunk + nb].astype(np.float32)
        T = min(a32.shape[1], ti.shape[1])
        a32, abf, gap, got = a32[:, :T], abf[:, :T], gap[:, :T], ti[:, :T, 0].astype(np.int64)
        valid = a32 >= 0
        conf = valid & (gap > 0.1)
        return dict(frames=int(valid.sum()), confident=int(conf.sum()),
                    engine=int((valid & (got != a32)).sum()), engine_confident=int((conf & (got != a32)).sum()),
                    ref_bf16=int((valid & (abf != a32)).sum()), ref_bf16_confident=int((conf & (abf != a32)).sum()))


def _assert_reduced_precision(name, dtype, ter, fm, ref, slack=1.0):
    """The engine's reduced-precision mode is held to the reference's own bf16 behaviour: token error rate against the fp32
    reference no worse than the reference-under-autocast's (+ margin; x `slack` for fp8), and on frames where the fp32
    reference decides with a margin > 0.1 it disagrees no more often than the reference-bf16 does (+ 0.05 % of the frames + 4)."""
    for m in MODES:
        bound = slack * ref.ter_bound(m)
        assert ter[m][0] <= bound * ter[m][1], f"{name} {dtype} {m}: TER {ter[m][0]}/{ter[m][1]} > {bound:.4f} (reference bf16 {ref.ter[m]:.4f})"
    lim = slack * (fm["ref_bf16_confident"] + 0.0005 * fm["confident"]) + 4      # + 4 frames: two-chunk cases have ~500 frames
    assert fm["engine_confident"] <= lim, f"{name} {dtype}: {fm}"


# ------------------------------------------------------------------------------------------------ small_66
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_small_66_chunks_two_slice_pipeline(dtype):
    case = LongCase("small_66")
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(len(case.js["lens"]))])
    lens = np.array(case.js["lens"], np.int32)
    eng = Engine(case.cfg, case.sd, dtype=dtype, device=0, max_chunks=len(lens), chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(x, lens, case.beam)                     # B = 66 >= 64: slices of 34 + 32 chunks, host search overlapped
    assert eng.encoder_lens().tolist() == case.js["encoder_lens"]
    m0 = _tap_metrics(eng, case, 0, 0)
    ref = RefBf16("small_66")
    fm = ref.frame_disagreement(eng)
    res = eng.search(MODES, case.ctc_weight, case.reverse_weight)
    ter = _ter(res, case)
    _record(case="small_66", dtype=dtype, chunks=len(lens), ter={m: list(v) for m, v in ter.items()}, chunk0=m0, frames=fm,
            reference_bf16_ter=ref.ter)
    if dtype == "f32":
        # exact-f32 MFMA: the reference's ids, chunk for chunk (and its CTC peak times)
        for m in MODES:
            assert ter[m][0] == 0, f"{m}: {ter[m][0]} token edits in {ter[m][2]} chunks of {len(lens)}"
        for r, g in zip(res["attention_rescoring"], case.golden("attention_rescoring")):
            assert list(r.times) == g["times"]
            assert abs(r.score - g["score"]) <= 2e-2 + 1e-4 * abs(g["score"])
        assert m0["cos"] > 0.999999 and m0["logp_max_abs"] < 2e-3
    else:
        assert m0["cos"] > BF16_COS and m0["logp_p99_abs"] <= BF16_LOGP_P99_ABS and m0["logp_mean_abs"] <= BF16_LOGP_MEAN_ABS, m0
        _assert_reduced_precision("small_66", dtype, ter, fm, ref)
    eng.close()


# bf16 token error rates are bounded by the REFERENCE's own bf16 behaviour (RefBf16 above; round 3), not by "measured +
# margin": reference under torch.autocast(cpu, bf16) vs its fp32 run -- small_66 greedy 3.5 % / rescored 4.6 %, r640_chunk
# 4/66 / 6/84, r640_1h (the bench workload) see tests/golden/r640_1h_refbf16.json; the engine (fp32 residual stream, fp32
# LayerNorm / softmax statistics) measured 1.5 % / 3.4 %, 4/66 / 6/84 and 4.7 % / 8.9 % in round 2.  The f32 mode has 0 edits.
# SURVEY.md 8d asks CTC log-probs within 5e-2 in bf16; measured on the frames whose argmax agrees with the reference:
# mean 0.009, 99th percentile 0.041, max 0.054 (r640).  Asserted: p99 <= 5e-2 and mean <= 2e-2; encoder cos-sim > 0.9999
# (measured 0.99998).
BF16_LOGP_P99_ABS, BF16_LOGP_MEAN_ABS, BF16_COS = 5e-2, 2e-2, 0.9999


# ------------------------------------------------------------------------------------------------ r640 (the bench model)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_r640_chunk_against_reference(dtype):
    case = LongCase("r640_chunk")
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(2)])
    lens = np.array(case.js["lens"], np.int32)
    eng = Engine(case.cfg, case.sd, dtype=dtype, device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(x, lens, case.beam)
    assert eng.encoder_lens().tolist() == case.js["encoder_lens"]
    m0 = _tap_metrics(eng, case, 0, 0)
    ref = RefBf16("r640_chunk")
    fm = ref.frame_disagreement(eng)
    res = eng.search(MODES, case.ctc_weight, case.reverse_weight)
    ter = _ter(res, case)
    _record(case="r640_chunk", dtype=dtype, ter={m: list(v) for m, v in ter.items()}, chunk0=m0, frames=fm, reference_bf16_ter=ref.ter)
    if dtype == "f32":
        for m in MODES:
            assert ter[m][0] == 0, f"{m}: {ter[m]}"
        assert list(res["attention_rescoring"][0].times) == case.golden("attention_rescoring")[0]["times"]
        assert m0["cos"] > 0.999999 and m0["enc_max_abs"] < 5e-3 and m0["logp_max_abs"] < 5e-3
    else:
        assert m0["cos"] > BF16_COS and m0["logp_p99_abs"] <= BF16_LOGP_P99_ABS and m0["logp_mean_abs"] <= BF16_LOGP_MEAN_ABS, m0
        _assert_reduced_precision("r640_chunk", dtype, ter, fm, ref)
    eng.close()


def test_r640_attention_with_keys_prefolded_by_the_qkv_gemm_equals_the_two_product_form(monkeypatch, lab):
    """Round 6: the bf16 encoder runs the rel-pos attention in its folded form, (q+u).(k+p) + (v-u).p, with K' = k + p written by
    the qkv GEMM's epilogue (GemmArgs::rowadd) and the second product a per-key table.  Against the two-product form of rounds
    1-5 (lab switch RVB_ATTN_PREFOLD=0) on the r640 chunk pair: the same mathematics with one rounding of k fewer and one
    accumulation chain instead of two -- encoder outputs agree far tighter than either agrees with the fp32 reference, the top-1
    CTC ids agree on all but a handful of near-tied frames, and the GEMM really wrote k + p (the K third of the qkv buffer differs
    between the two runs by exactly the positional rows)."""
    case = LongCase("r640_chunk")
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(2)])
    lens = np.array(case.js["lens"], np.int32)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVB_ATTN_PREFOLD", flag)
        eng = Engine(case.cfg, case.sd, dtype="bf16", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
        eng.encode(x, lens, case.beam)
        _, ti = eng.ctc_topk()
        out[flag] = (eng.encoder_out().astype(np.float64), ti[:, :, 0].copy(), _tap_metrics(eng, case, 0, 0))
        eng.close()
    (e0, i0, m0), (e1, i1, m1) = out["0"], out["1"]
    n = case.js["encoder_lens"]
    cos = min(float((e0[b, :n[b]] * e1[b, :n[b]]).sum() / (np.linalg.norm(e0[b, :n[b]]) * np.linalg.norm(e1[b, :n[b]]))) for b in range(2))
    agree = float(np.mean([np.mean(i0[b, :n[b]] == i1[b, :n[b]]) for b in range(2)]))
    _record(case="r640_chunk", test="attention_prefold_vs_two_products", cos=cos, top1_agree=agree, two_products=m0, prefolded=m1)
    assert cos > 0.99995 and agree > 0.98, (cos, agree)
    assert m1["cos"] > BF16_COS and m1["logp_p99_abs"] <= BF16_LOGP_P99_ABS, m1


def test_r640_glu_in_the_pointwise_gemm_equals_the_two_kernel_form(monkeypatch, lab):
    """Round 6: pointwise_conv1 + GLU of the convolution module in ONE kernel -- the GEMM runs on the interleaved copy of the weights
    and its epilogue stores a * sigmoid(b) (ACT_GLU) -- against GEMM + gate-while-staging (lab switch RVB_GLU_FUSE=0).  The fused form
    gates the fp32 accumulators and rounds once; the other rounds a and b to bf16 first: encoder outputs agree far tighter than either
    agrees with the fp32 reference."""
    case = LongCase("r640_chunk")
    x = np.concatenate([case.chunk_feats(c)[0] for c in range(2)])
    lens = np.array(case.js["lens"], np.int32)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RVB_GLU_FUSE", flag)
        eng = Engine(case.cfg, case.sd, dtype="bf16", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
        eng.encode(x, lens, case.beam)
        _, ti = eng.ctc_topk()
        out[flag] = (eng.encoder_out().astype(np.float64), ti[:, :, 0].copy(), _tap_metrics(eng, case, 0, 0))
        eng.close()
    (e0, i0, m0), (e1, i1, m1) = out["0"], out["1"]
    n = case.js["encoder_lens"]
    cos = min(float((e0[b, :n[b]] * e1[b, :n[b]]).sum() / (np.linalg.norm(e0[b, :n[b]]) * np.linalg.norm(e1[b, :n[b]]))) for b in range(2))
    agree = float(np.mean([np.mean(i0[b, :n[b]] == i1[b, :n[b]]) for b in range(2)]))
    _record(case="r640_chunk", test="glu_fused_vs_two_kernels", cos=cos, top1_agree=agree, two_kernels=m0, fused=m1)
    assert not np.array_equal(e0, e1)                     # it really is the other path
    assert cos > 0.99995 and agree > 0.98, (cos, agree)
    assert m1["cos"] > BF16_COS and m1["logp_p99_abs"] <= BF16_LOGP_P99_ABS, m1
    assert m1["cos"] >= m0["cos"] - 2e-6, (m0["cos"], m1["cos"])      # one rounding fewer: not further from the fp32 reference


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_r640_one_hour_bench_workload_against_reference(dtype):
    """bench.py's step, checked: PCM -> device fbank -> decode_resident(176 chunks: slices of 144 + 32) -> attention
    rescoring, against the reference's tokens for the same hour of audio."""
    case = LongCase("r640_1h")
    n = len(case.js["lens"])
    eng = Engine(case.cfg, case.sd, dtype=dtype, device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.upload_pcm(case.pcm)
    nf = eng.fbank()
    assert nf == sum(case.js["lens"])
    res = eng.decode_resident(nf, MODES, case.chunk, case.beam, case.ctc_weight, case.reverse_weight)
    assert len(res["attention_rescoring"]) == n == 176
    ter = _ter(res, case)
    m_last = _tap_metrics(eng, case, n - 1, n - 1)        # the last chunk sits in the 32-chunk slice
    ref = RefBf16("r640_1h")
    fm = ref.frame_disagreement(eng)
    _record(case="r640_1h", dtype=dtype, chunks=n, ter={m: list(v) for m, v in ter.items()}, last_chunk=m_last, frames=fm,
            reference_bf16_ter=ref.ter)
    if dtype == "f32":
        # the device fbank differs from the oracle features by ~1e-4 (fp32 FFT order): a near-tied frame may flip
        for m in MODES:
            assert ter[m][0] <= 0.002 * ter[m][1], f"{m}: {ter[m]}"
        assert m_last["cos"] > 0.99999
    else:
        assert m_last["cos"] > BF16_COS and m_last["logp_p99_abs"] <= BF16_LOGP_P99_ABS and m_last["logp_mean_abs"] <= BF16_LOGP_MEAN_ABS, m_last
        _assert_reduced_precision("r640_1h", dtype, ter, fm, ref)
    eng.close()
