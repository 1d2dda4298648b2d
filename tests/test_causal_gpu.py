"""Causal convolution module + cnn cache on the GPU (`encoder_conf.causal: true`; transformer/convolution.py:55-57,
113-121, encoder.py:231-341) against goldens of the UNMODIFIED reference (oracle/gen_golden_causal.py): offline decoding
of a padded batch with and without chunk masks, forward_chunk_by_chunk (incl. chunks shorter than the cnn cache) and
ASRModel.decode(simulate_streaming=True)."""
import numpy as np
import pytest

from golden_util import MODES, CausalCase
from reverb_amd.engine import Engine
from util import token_error_rate

pytestmark = pytest.mark.gpu


def _ter(got, want):
    return token_error_rate(got, want)


def _within_reference_bf16(got, want, ref_edits, what):
    """Round 4: the bf16 bound is the UNMODIFIED reference's own behaviour under torch.autocast('cpu', bfloat16) on the same
    inputs (oracle/gen_golden_bf16ref_short.py -> tests/golden/short_refbf16.json), + 1 % of the tokens rounded up."""
    import math
    from util import edit_distance
    err = sum(edit_distance(g, w) for g, w in zip(got, want))
    tot = sum(len(w) for w in want)
    assert tot == ref_edits[1], (what, tot, ref_edits)
    assert err <= ref_edits[0] + math.ceil(0.01 * tot), f"{what}: {err}/{tot} token edits, reference-bf16 {ref_edits}"


def _short_refbf16():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "short_refbf16.json")) as f:
        return json.load(f)["cases"]["tiny_causal"]


def _check_rows(res, rows, tag):
    for b, want in enumerate(rows):
        assert list(res["ctc_greedy_search"][b].tokens) == want["greedy"], tag
        p = res["ctc_prefix_beam_search"][b]
        assert list(p.tokens) == want["prefix"] and list(p.times) == want["prefix_times"], tag
        assert [list(h) for h in p.nbest] == want["nbest"], tag
        np.testing.assert_allclose(p.nbest_scores, want["nbest_scores"], rtol=0, atol=2e-2)
        r = res["attention_rescoring"][b]
        assert list(r.tokens) == want["rescoring"] and list(r.times) == want["rescoring_times"], tag
        assert abs(r.score - want["rescoring_score"]) <= 2e-2, tag


def test_causal_offline_f32_matches_reference_golden():
    case = CausalCase("tiny_causal")
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    for run in case.js["offline"]:
        cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
        eng.apply_decoding_chunk(cs, left)
        eng.encode(case.x, case.lens, case.beam)
        assert eng.encoder_lens().tolist() == run["encoder_lens"]
        enc = eng.encoder_out()[:, ::4]
        want = case.arrays[f"enc_{cs}_{left}".replace("-", "m")]
        for b, n in enumerate(run["encoder_lens"]):
            nn = len(range(0, n, 4))
            np.testing.assert_allclose(enc[b, :nn], want[b, :nn], rtol=2e-3, atol=2e-3)
        _check_rows(eng.search(MODES, case.ctc_weight, case.reverse_weight), run["chunks"], (cs, left))
    eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_causal_forward_chunk_by_chunk_matches_reference(dtype):
    case = CausalCase("tiny_causal")
    eng = Engine(case.cfg, case.sd, dtype=dtype, device=0, max_chunks=2, chunk_frames=case.chunk, cat_embs=case.cat)
    greedy_got, greedy_want = [], []
    for run in case.js["streaming"]:
        cs, left = run["decoding_chunk_size"], run["num_decoding_left_chunks"]
        ys = eng.forward_chunk_by_chunk(case.feats, cs, left)
        assert ys.shape[0] == run["out_frames"]
        assert eng.stream_state() == (run["out_frames"], run["final_cache_frames"])
        want = case.arrays[f"ys_{cs}_{left}".replace("-", "m")]
        if dtype == "f32":
            np.testing.assert_allclose(ys[::4], want, rtol=2e-3, atol=2e-3)
        else:
            a, g = ys[::4].ravel().astype(np.float64), want.ravel().astype(np.float64)
            assert a @ g / (np.linalg.norm(a) * np.linalg.norm(g)) > 0.999
        eng.stream_finish(case.beam)
        got = eng.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"][0]
        if dtype == "f32":
            assert list(got.tokens) == run["greedy"], (cs, left)
        greedy_got.append(list(got.tokens)); greedy_want.append(run["greedy"])
    if dtype == "bf16":
        _within_reference_bf16(greedy_got, greedy_want, _short_refbf16()["streaming_total"], "forward_chunk_by_chunk, five settings")
    # the offline path right after a stream, and a fresh stream after it, see no stale cnn cache
    eng.apply_decoding_chunk(-1, -1)
    eng.encode(case.x, case.lens, case.beam)
    off = [list(r.tokens) for r in eng.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"]]
    run = case.js["streaming"][3]
    ys2 = eng.forward_chunk_by_chunk(case.feats, run["decoding_chunk_size"], run["num_decoding_left_chunks"])
    if dtype == "f32":
        assert off == [c["greedy"] for c in case.js["offline"][0]["chunks"]]
        np.testing.assert_allclose(ys2[::4], case.arrays["ys_3_0"], rtol=2e-3, atol=2e-3)
    eng.close()


def test_causal_decode_simulate_streaming_matches_reference():
    """Even kernel (K = 8) + BatchNorm, no language-specific layers: ASRModel.decode(simulate_streaming=True) end to end and
    the offline path of the same model, f32, token-exact."""
    import torch
    from reverb_amd.reverb import RvbASRModel
    case = CausalCase("tiny_causal_plain")
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=2, chunk_frames=case.chunk)
    model = RvbASRModel(eng)
    x, lens = torch.from_numpy(case.x), torch.from_numpy(case.lens)
    res = model.decode(MODES, x, lens, case.beam, -1, -1, case.ctc_weight, False, case.reverse_weight)
    _check_rows(res, case.js["offline"][0]["chunks"], "offline")
    for run in case.js["streaming"]:
        res = model.decode(MODES, x, lens, case.beam, run["decoding_chunk_size"], run["num_decoding_left_chunks"], case.ctc_weight,
                           True, case.reverse_weight)
        _check_rows(res, run["chunks"], (run["decoding_chunk_size"], run["num_decoding_left_chunks"]))
    eng.close()


def test_causal_bf16_offline_close_to_reference():
    case = CausalCase("tiny_causal")
    eng = Engine(case.cfg, case.sd, dtype="bf16", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    run = case.js["offline"][0]
    eng.apply_decoding_chunk(-1, -1)
    eng.encode(case.x, case.lens, case.beam)
    enc = eng.encoder_out()[:, ::4]
    want = case.arrays["enc_m1_m1"]
    for b, n in enumerate(run["encoder_lens"]):
        nn = len(range(0, n, 4))
        a, g = enc[b, :nn].ravel().astype(np.float64), want[b, :nn].ravel().astype(np.float64)
        assert a @ g / (np.linalg.norm(a) * np.linalg.norm(g)) > 0.999
    res = eng.search(["ctc_greedy_search"], 0.0, 0.0)["ctc_greedy_search"]
    _within_reference_bf16([list(r.tokens) for r in res], [c["greedy"] for c in run["chunks"]], _short_refbf16()["offline"]["-1_-1"]["edits"],
                           "offline greedy")
    eng.close()
