"""Streaming encoder on the GPU (rvb_stream_begin / rvb_stream_chunk / rvb_stream_finish) against the unmodified
reference's forward_chunk_by_chunk (asr/wenet/transformer/encoder.py:231-402) and ASRModel.decode(simulate_streaming=True)
(asr_model.py:301-306); goldens by oracle/gen_golden_streaming.py."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, Case
from oracle import fbank_ref
from reverb_amd import synth
from reverb_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_chunk_by_chunk_matches_reference(dtype):
    case = Case("tiny_ln")
    with open(os.path.join(GOLDEN, "tiny_ln_streaming.json")) as f:
        gold = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "tiny_ln_streaming.npz"))
    feats = fbank_ref.fbank(case.pcm)
    eng = Engine(case.cfg, case.sd, dtype=dtype, device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    for run in gold["runs"]:
        cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
        ys = eng.forward_chunk_by_chunk(feats, cs, left)
        assert ys.shape[0] == run["out_frames"]
        assert eng.stream_state() == (run["out_frames"], run["final_cache_frames"])
        want = arrays[f"ys_{cs}_{left}".replace("-", "m")]
        if dtype == "f32":
            np.testing.assert_allclose(ys[::4], want, rtol=2e-3, atol=2e-3)
        else:
            a, g = ys[::4].ravel().astype(np.float64), want.ravel().astype(np.float64)
            assert a @ g / (np.linalg.norm(a) * np.linalg.norm(g)) > 0.999
        eng.stream_finish(case.beam)
        assert eng.encoder_lens().tolist() == [run["out_frames"]]
        np.testing.assert_array_equal(eng.encoder_out()[0], ys)            # the finished stream is the current batch
        got = eng.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"][0]
        if dtype == "f32":
            assert list(got.tokens) == run["greedy"], (cs, left)
    # a stream is independent of what the engine did before: offline encode in between, then the first setting again
    x, lens = case.chunked_feats()
    eng.encode(x, lens, case.beam)
    offline = [list(r.tokens) for r in eng.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"]]
    if dtype == "f32":
        assert offline == [g["tokens"] for g in case.golden("ctc_greedy_search")]
    run = gold["runs"][0]
    ys2 = eng.forward_chunk_by_chunk(feats, run["decoding_chunk_size"], run["num_decoding_left_chunks"])
    if dtype == "f32":
        np.testing.assert_allclose(ys2[::4], arrays["ys_16_m1"], rtol=2e-3, atol=2e-3)
    eng.close()


def test_forward_chunk_single_calls_and_limits():
    from reverb_amd._lib import RvbError
    case = Case("tiny_ln")
    feats = fbank_ref.fbank(case.pcm)
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    with pytest.raises(RvbError, match="rvb_stream_begin"):
        eng.forward_chunk(feats[:67])
    eng.stream_begin()
    y0 = eng.forward_chunk(feats[:67], 16)               # 16 frames out, cache keeps 16
    assert y0.shape == (16, case.cfg["encoder_conf"]["output_size"]) and eng.stream_state() == (16, 16)
    y1 = eng.forward_chunk(feats[64:64 + 35], 16)        # a shorter chunk: 8 frames; cache = last 16 of 24
    assert y1.shape[0] == 8 and eng.stream_state() == (24, 16)
    with pytest.raises(RvbError, match="7 input frames"):
        eng.forward_chunk(feats[:6])
    with pytest.raises(RvbError, match="5000|positional"):
        for _ in range(400):
            eng.forward_chunk(feats[:67], 0, return_output=False)
    eng.close()


def test_decode_simulate_streaming_matches_reference():
    """ASRModel.decode(..., simulate_streaming=True) end to end on a model without language-specific layers (the only kind
    the reference's seam can run): greedy, prefix beam n-best and rescoring per chunk, f32, token-exact."""
    import torch
    from reverb_amd.reverb import RvbASRModel
    with open(os.path.join(GOLDEN, "tiny_plain_streaming.json")) as f:
        gold = json.load(f)
    c = gold["case"]
    cfg = synth.make_config(c["dims"], c["norm"])
    cfg["dataset_conf"]["pass_cat_emb"] = False
    sd = synth.make_state_dict(cfg, c["seed"], gold["gamma"], gold["beta"])
    assert not any("language_layers" in k for k in sd)
    feats = fbank_ref.fbank(synth.synth_audio(c["seconds"], seed=1234 + c["seed"]))
    nch = -(-feats.shape[0] // c["chunk"])
    x = np.zeros((nch, c["chunk"], 80), np.float32)
    lens = np.zeros(nch, np.int32)
    for i in range(nch):
        part = feats[i * c["chunk"]:(i + 1) * c["chunk"]]
        x[i, :len(part)] = part
        lens[i] = len(part)
    assert lens.tolist() == gold["lens"]
    eng = Engine(cfg, sd, dtype="f32", device=0, max_chunks=2, chunk_frames=c["chunk"])
    model = RvbASRModel(eng)
    modes = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]
    for run in gold["runs"]:
        res = model.decode(modes, torch.from_numpy(x), torch.from_numpy(lens), c["beam"], run["decoding_chunk_size"],
                           run["num_decoding_left_chunks"], c["ctc_weight"], True, c["reverse_weight"])
        for b, want in enumerate(run["chunks"]):
            assert list(res["ctc_greedy_search"][b].tokens) == want["greedy"], (run["decoding_chunk_size"], b)
            p = res["ctc_prefix_beam_search"][b]
            assert list(p.tokens) == want["prefix"] and list(p.times) == want["prefix_times"]
            assert [list(h) for h in p.nbest] == want["nbest"]
            r = res["attention_rescoring"][b]
            assert list(r.tokens) == want["rescoring"] and list(r.times) == want["rescoring_times"]
            assert abs(r.score - want["rescoring_score"]) <= 2e-2
    eng.close()
