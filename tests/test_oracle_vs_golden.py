"""Pins the oracle (oracle/model_ref.py, search_ref.py, fbank_ref.py) against the golden vectors
that oracle/gen_golden.py produced by running the unmodified reference.  CPU only."""
import numpy as np
import pytest
import torch

from golden_util import CASES, MODES, Case, CausalCase, GOLDEN, JointCase, LongCase
from oracle import fbank_ref, model_ref as M, search_ref as S


@pytest.mark.parametrize("name", [c for c in CASES if c != "r268_chunk"])
def test_decode_matches_reference_golden(name):
    case = Case(name)
    x, lens = case.chunked_feats()
    sd = M.to_torch_sd(case.sd)
    taps = {}
    res = S.decode(sd, case.cfg, MODES, torch.from_numpy(x), torch.from_numpy(lens), case.beam,
                   ctc_weight=case.ctc_weight, reverse_weight=case.reverse_weight, cat_embs=torch.tensor(case.cat), taps=taps)
    assert taps["encoder_lens"].tolist() == case.js["encoder_lens"]
    # same torch ops as the reference => bit-identical encoder output
    np.testing.assert_array_equal(taps["encoder_out"].numpy(), case.arrays["encoder_out"])
    tv, ti = taps["ctc_probs"].topk(case.beam, dim=-1)
    np.testing.assert_array_equal(ti.numpy(), case.arrays["topk_idx"])
    np.testing.assert_array_equal(tv.numpy(), case.arrays["topk_val"])
    for mode in MODES:
        for got, want in zip(res[mode], case.golden(mode)):
            assert list(got.tokens) == want["tokens"], (name, mode)
            if want["times"] is not None:
                assert list(got.times) == want["times"]
            if want["nbest"] is not None:
                assert [list(h) for h in got.nbest] == want["nbest"]
                assert got.nbest_times == want["nbest_times"]
                assert got.nbest_scores == want["nbest_scores"]            # float64 bit-exact
            if mode == "attention_rescoring":
                assert float(got.score) == want["score"]
                assert got.confidence == want["confidence"]
                assert got.tokens_confidence == want["tokens_confidence"]


def test_r268_first_chunk_tokens():
    """d=640 / 8 heads (head dim 80) planning point: one full chunk through the oracle."""
    case = Case("r268_chunk")
    x, lens = case.chunked_feats()
    sd = M.to_torch_sd(case.sd)
    res = S.decode(sd, case.cfg, ["ctc_greedy_search"], torch.from_numpy(x[:1]), torch.from_numpy(lens[:1]), case.beam,
                   cat_embs=torch.tensor(case.cat))
    assert list(res["ctc_greedy_search"][0].tokens) == case.golden("ctc_greedy_search")[0]["tokens"]


def test_fbank_restatement_vs_independent_implementation():
    """The reference's fbank lives in torchaudio (absent): parity unpinned.  Cross-check the
    restatement against the Kaldi-compatible implementation in `transformers` (committed golden)."""
    from reverb_amd import synth
    g = np.load(GOLDEN + "/fbank_transformers.npz")["feats"]
    f = fbank_ref.fbank(synth.synth_audio(2.0, seed=99))
    assert f.shape == g.shape
    np.testing.assert_allclose(f, g, rtol=0, atol=2e-4)


def test_fbank_edges():
    assert fbank_ref.fbank(np.zeros(399, np.int16)).shape == (0, 80)
    assert fbank_ref.fbank(np.zeros(400, np.int16)).shape == (1, 80)
    assert fbank_ref.fbank(np.zeros(559, np.int16)).shape == (1, 80)
    assert fbank_ref.fbank(np.zeros(560, np.int16)).shape == (2, 80)
    # silence -> log of the floor
    np.testing.assert_allclose(fbank_ref.fbank(np.zeros(400, np.int16)), np.log(np.float32(1.1920929e-07)), rtol=1e-6)


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_ln_r2l", "tiny_bn", "small_ln"])
def test_attention_mode_oracle_matches_reference_golden(name):
    """`attention` mode (search.py:251-360): the oracle's K/V-cached restatement reproduces the tokens the
    unmodified reference (cached forward_one_step) produced -- oracle/gen_golden_attention.py."""
    import json
    case = Case(name)
    with open(GOLDEN + f"/{name}_attention.json") as f:
        gold = json.load(f)
    x, lens = case.chunked_feats()
    sd = M.to_torch_sd(case.sd)
    with torch.no_grad():
        enc, mask = M.encoder_forward(sd, case.cfg, torch.from_numpy(x), torch.from_numpy(lens), torch.tensor(case.cat))
        for run in gold["runs"]:
            res = S.attention_beam_search(sd, case.cfg, enc, mask, case.beam, run["length_penalty"], torch.tensor(case.cat))
            assert [list(r.tokens) for r in res] == run["tokens"], (name, run["length_penalty"])


def test_chunk_masked_encoder_oracle_matches_reference_golden():
    """decoding_chunk_size / num_decoding_left_chunks on a use_dynamic_chunk model (utils/mask.py:86-197): the oracle's
    encoder reproduces the unmodified reference bit for bit (oracle/gen_golden_chunkmask.py)."""
    import json
    case = Case("tiny_ln")
    with open(GOLDEN + "/tiny_ln_chunkmask.json") as f:
        gold = json.load(f)
    arrays = np.load(GOLDEN + "/tiny_ln_chunkmask.npz")
    cfg = dict(case.cfg)
    cfg["encoder_conf"] = dict(cfg["encoder_conf"], use_dynamic_chunk=True)
    x, lens = case.chunked_feats()
    sd = M.to_torch_sd(case.sd)
    with torch.no_grad():
        for run in gold["runs"]:
            cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
            enc, mask = M.encoder_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(lens), torch.tensor(case.cat),
                                          decoding_chunk_size=cs, num_decoding_left_chunks=left)
            np.testing.assert_array_equal(enc.numpy()[:, ::4], arrays[f"enc_{cs}_{left}".replace("-", "m")])
            probs = M.ctc_logprobs(sd, enc)
            got = S.ctc_greedy_search(probs, mask.squeeze(1).sum(1), 0)
            assert [list(r.tokens) for r in got] == run["greedy"], (cs, left)
    # a model without use_dynamic_chunk ignores the arguments (full context)
    with torch.no_grad():
        a, _ = M.encoder_forward(sd, case.cfg, torch.from_numpy(x[:1]), torch.from_numpy(lens[:1]), torch.tensor(case.cat), decoding_chunk_size=16)
        b, _ = M.encoder_forward(sd, case.cfg, torch.from_numpy(x[:1]), torch.from_numpy(lens[:1]), torch.tensor(case.cat))
    assert torch.equal(a, b)


@pytest.mark.parametrize("name,chunks", [("small_66", (0, 1, 33, 64, 65)), ("r640_chunk", (0, 1)), ("r640_1h", (100,))])
def test_oracle_matches_long_form_reference_golden(name, chunks):
    """Long-form goldens (the reference decoded chunk by chunk, batch 1): the oracle reproduces the reference's greedy and
    rescored winners on sampled chunks -- including one chunk of the 1 h r640 recording bench.py times, whose weights are
    the frozen-beta ones (`synth.calibrated_state_dict`)."""
    from reverb_amd import synth
    case = LongCase(name)
    if case.c.get("frozen_beta"):
        assert case.js["beta"] == synth.CTC_BLANK_BIAS[(case.c["dims"], case.c["seed"])]
    sd = M.to_torch_sd(case.sd)
    for c in chunks:
        x, lens = case.chunk_feats(c)
        taps = {}
        res = S.decode(sd, case.cfg, ["ctc_greedy_search", "attention_rescoring"], torch.from_numpy(x), torch.from_numpy(lens),
                       case.beam, ctc_weight=case.ctc_weight, reverse_weight=case.reverse_weight, cat_embs=torch.tensor(case.cat), taps=taps)
        assert taps["encoder_lens"].tolist() == [case.js["encoder_lens"][c]]
        assert list(res["ctc_greedy_search"][0].tokens) == case.golden("ctc_greedy_search")[c]["tokens"], (name, c)
        g = case.golden("attention_rescoring")[c]
        r = res["attention_rescoring"][0]
        assert list(r.tokens) == g["tokens"] and list(r.times) == g["times"], (name, c)
        assert float(r.score) == g["score"] and r.confidence == g["confidence"]
        key = f"encoder_out_{c}"
        if key in case.arrays:
            n = case.js["encoder_lens"][c]
            # (bit-identical on full chunks; a 1-frame tail chunk takes another torch GEMM path in nn.Linear than in F.linear)
            np.testing.assert_allclose(taps["encoder_out"][0, :n:16, ::8].numpy(), case.arrays[key], rtol=0, atol=2e-5)


def test_streaming_encoder_oracle_matches_reference_golden():
    """forward_chunk / forward_chunk_by_chunk with attention caches (encoder.py:231-402): the oracle's restatement against
    the unmodified reference on the language-specific tiny model (oracle/gen_golden_streaming.py)."""
    import json
    case = Case("tiny_ln")
    with open(GOLDEN + "/tiny_ln_streaming.json") as f:
        gold = json.load(f)
    arrays = np.load(GOLDEN + "/tiny_ln_streaming.npz")
    feats = torch.from_numpy(fbank_ref.fbank(case.pcm)).unsqueeze(0)
    assert feats.shape[1] == gold["frames"]
    sd = M.to_torch_sd(case.sd)
    with torch.no_grad():
        for run in gold["runs"]:
            cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
            ys, cache_frames = M.encoder_forward_chunk_by_chunk(sd, case.cfg, feats, cs, left, torch.tensor(case.cat))
            assert ys.shape[1] == run["out_frames"] and cache_frames == run["final_cache_frames"], (cs, left)
            np.testing.assert_allclose(ys[0, ::4].numpy(), arrays[f"ys_{cs}_{left}".replace("-", "m")], rtol=0, atol=2e-5)
            got = S.ctc_greedy_search(M.ctc_logprobs(sd, ys), torch.tensor([ys.shape[1]]), 0)
            assert list(got[0].tokens) == run["greedy"], (cs, left)


def _check_rows(res, rows, tag):
    for b, want in enumerate(rows):
        assert list(res["ctc_greedy_search"][b].tokens) == want["greedy"], tag
        p = res["ctc_prefix_beam_search"][b]
        assert list(p.tokens) == want["prefix"] and list(p.times) == want["prefix_times"], tag
        assert [list(h) for h in p.nbest] == want["nbest"], tag
        np.testing.assert_allclose(p.nbest_scores, want["nbest_scores"], rtol=0, atol=1e-4)
        r = res["attention_rescoring"][b]
        assert list(r.tokens) == want["rescoring"] and list(r.times) == want["rescoring_times"], tag
        assert abs(float(r.score) - want["rescoring_score"]) < 1e-3, tag


def test_causal_model_oracle_matches_reference_golden():
    """encoder_conf.causal (convolution.py:55-57,113-121): the oracle's left-padded convolution module and its cnn cache
    against the unmodified reference (oracle/gen_golden_causal.py) -- offline decode of a padded batch with chunk masks,
    forward_chunk_by_chunk incl. chunks shorter than the cache, and the final cnn cache itself."""
    case = CausalCase("tiny_causal")
    sd = M.to_torch_sd(case.sd)
    cat = torch.tensor(case.cat)
    x, lens = torch.from_numpy(case.x), torch.from_numpy(case.lens)
    for run in case.js["offline"]:
        cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
        taps = {}
        res = S.decode(sd, case.cfg, MODES, x, lens, case.beam, ctc_weight=case.ctc_weight, reverse_weight=case.reverse_weight,
                       cat_embs=cat, taps=taps, decoding_chunk_size=cs, num_decoding_left_chunks=left)
        assert taps["encoder_lens"].tolist() == run["encoder_lens"]
        np.testing.assert_array_equal(taps["encoder_out"].numpy()[:, ::4], case.arrays[f"enc_{cs}_{left}".replace("-", "m")])
        _check_rows(res, run["chunks"], (cs, left))
    feats = torch.from_numpy(case.feats).unsqueeze(0)
    with torch.no_grad():
        for run in case.js["streaming"]:
            cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
            key = f"{cs}_{left}".replace("-", "m")
            ys, cache_frames, cnn = M.encoder_forward_chunk_by_chunk(sd, case.cfg, feats, cs, left, cat, return_cnn_cache=True)
            assert ys.shape[1] == run["out_frames"] and cache_frames == run["final_cache_frames"], (cs, left)
            np.testing.assert_allclose(ys[0, ::4].numpy(), case.arrays["ys_" + key], rtol=0, atol=2e-5)
            assert list(cnn.shape) == run["cnn_cache_shape"]
            np.testing.assert_allclose(cnn.numpy(), case.arrays["cnn_" + key], rtol=0, atol=2e-5)
            got = S.ctc_greedy_search(M.ctc_logprobs(sd, ys), torch.tensor([ys.shape[1]]), 0)
            assert list(got[0].tokens) == run["greedy"], (cs, left)


def test_causal_plain_model_oracle_matches_reference_golden():
    """Even depthwise kernel (K = 8, legal only when causal) + BatchNorm, no language-specific layers: offline decode and
    the simulate_streaming seam (asr_model.py:301-306: forward_chunk_by_chunk of the whole padded input, lengths ignored)."""
    case = CausalCase("tiny_causal_plain")
    sd = M.to_torch_sd(case.sd)
    x, lens = torch.from_numpy(case.x), torch.from_numpy(case.lens)
    res = S.decode(sd, case.cfg, MODES, x, lens, case.beam, ctc_weight=case.ctc_weight, reverse_weight=case.reverse_weight)
    _check_rows(res, case.js["offline"][0]["chunks"], "offline")
    with torch.no_grad():
        for run in case.js["streaming"]:
            cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
            for b, want in enumerate(run["chunks"]):
                ys, _ = M.encoder_forward_chunk_by_chunk(sd, case.cfg, x[b:b + 1], cs, left, None)
                n = torch.tensor([ys.shape[1]])
                probs = M.ctc_logprobs(sd, ys)
                assert list(S.ctc_greedy_search(probs, n, 0)[0].tokens) == want["greedy"], (cs, left, b)
                pref = S.ctc_prefix_beam_search(probs, n, case.beam, 0)
                assert [list(h) for h in pref[0].nbest] == want["nbest"], (cs, left, b)
                r = S.attention_rescoring(sd, case.cfg, pref, ys, n, case.ctc_weight, case.reverse_weight, None)[0]
                assert list(r.tokens) == want["rescoring"] and abs(float(r.score) - want["rescoring_score"]) < 1e-3


@pytest.mark.parametrize("name", ["joint_tiny", "joint_small"])
def test_joint_decoding_oracle_matches_reference_class(name):
    """`joint_decoding` (search.py:450-496; espnet/beam_search_timesync.py): the oracle's restatement against the reference's
    own BeamSearchTimeSync class driven with (1, len, d) memories (oracle/gen_golden_joint.py) -- winner, start / end frames,
    per-token confidences, joint score.  The golden also records that the reference's entry point itself raises."""
    case = JointCase(name)
    assert case.js["reference_entry_point"].startswith("RuntimeError")
    sd = M.to_torch_sd(case.sd)
    cat = torch.tensor(case.cat)
    with torch.no_grad():
        enc, mask = M.encoder_forward(sd, case.cfg, torch.from_numpy(case.x), torch.from_numpy(case.lens), cat)
        probs = M.ctc_logprobs(sd, enc)
    lens = mask.squeeze(1).sum(1)
    assert lens.tolist() == case.js["encoder_lens"]
    for run in case.js["runs"]:
        bp = run.get("blank_penalty", 0.0)
        with torch.no_grad():
            lp = M.ctc_logprobs(sd, enc, bp, 0) if bp else probs
        got = S.joint_decoding(sd, case.cfg, enc, lens, lp, run["ctc_weight"], run["beam"], run["pre_beam_ratio"],
                               run["length_bonus"], cat)
        for b, want in enumerate(run["chunks"]):
            g = got[b]
            assert list(g.tokens) == want["tokens"], (run, b)
            assert list(g.times) == want["times"] and list(g.end_times) == want["end_times"], (run, b)
            assert abs(g.score - want["score"]) < 2e-3 * max(1.0, abs(want["score"])), (run, b, g.score, want["score"])
            np.testing.assert_allclose(g.tokens_confidence, want["tokens_confidence"], rtol=2e-3, atol=1e-6)
