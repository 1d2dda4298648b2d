"""`joint_decoding` on the GPU (rvb_joint_decode: every chunk of the batch advances one encoder frame per iteration, one
batched decoder step for the new prefixes of all chunks) against goldens of the reference's own BeamSearchTimeSync class
(oracle/gen_golden_joint.py; asr/wenet/transformer/search.py:450-496, espnet/beam_search_timesync.py)."""
import numpy as np
import pytest

from golden_util import JointCase
from reverb_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["joint_tiny", "joint_small"])
def test_joint_decoding_f32_matches_reference_class(name):
    case = JointCase(name)
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    for run in case.js["runs"]:
        beam = run["beam"]
        # the pre-beam + 8 log-probs per frame, as ASRModel.decode asks for (engine.joint_topk); round 4: pre-beams beyond 16
        # (beam 12 -> 18 candidates) and a blank penalty (the reference's ctc_logprobs applies it before the mode sees the rows)
        eng.encode(case.x, case.lens, beam, run.get("blank_penalty", 0.0), topk=min(int(run["pre_beam_ratio"] * beam) + 8, 64))
        assert eng.encoder_lens().tolist() == case.js["encoder_lens"]
        got = eng.joint_decode(run["ctc_weight"], run["length_bonus"], run["pre_beam_ratio"])
        rows, steps = eng.joint_stats()
        assert steps <= 1 + max(case.js["encoder_lens"]) and rows >= len(case.lens)      # one batched decoder step per frame at most
        for b, want in enumerate(run["chunks"]):
            g = got[b]
            assert list(g.tokens) == want["tokens"], (run, b)
            assert list(g.times) == want["times"] and list(g.end_times) == want["end_times"], (run, b)
            assert abs(g.score - want["score"]) <= 2e-2 + 1e-3 * abs(want["score"]), (run, b, g.score, want["score"])
            np.testing.assert_allclose(g.tokens_confidence, want["tokens_confidence"], rtol=2e-2, atol=1e-5)
    eng.close()


def test_joint_decoding_through_decode_seam_and_with_other_modes():
    """ASRModel.decode(['joint_decoding', ...]): the mode next to the prefix beam and rescoring in one call (the encoder keeps
    int(1.5 * beam) log-probs per frame for the pre-beam, the other modes use the first `beam` of them) -- results equal to
    the modes run alone."""
    import torch
    from reverb_amd.reverb import RvbASRModel
    case = JointCase("joint_tiny")
    run = case.js["runs"][0]
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    model = RvbASRModel(eng)
    x, lens = torch.from_numpy(case.x), torch.from_numpy(case.lens)
    both = model.decode(["joint_decoding", "ctc_prefix_beam_search", "attention_rescoring"], x, lens, run["beam"], ctc_weight=run["ctc_weight"],
                        length_penalty=run["length_bonus"])
    alone = model.decode(["ctc_prefix_beam_search", "attention_rescoring"], x, lens, run["beam"], ctc_weight=run["ctc_weight"])
    for b, want in enumerate(run["chunks"]):
        assert list(both["joint_decoding"][b].tokens) == want["tokens"]
        for m in ("ctc_prefix_beam_search", "attention_rescoring"):
            assert list(both[m][b].tokens) == list(alone[m][b].tokens) and both[m][b].times == alone[m][b].times
    # a bf16 run decodes something of the same kind (no exactness claim: near-tied hypotheses)
    eng.close()
    eng = Engine(case.cfg, case.sd, dtype="bf16", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    run = case.js["runs"][2]
    eng.encode(case.x, case.lens, run["beam"], topk=int(run["pre_beam_ratio"] * run["beam"]))
    got = eng.joint_decode(run["ctc_weight"], run["length_bonus"], run["pre_beam_ratio"])
    for b, want in enumerate(run["chunks"]):
        assert abs(len(got[b].tokens) - len(want["tokens"])) <= max(4, len(want["tokens"]) // 4)
    eng.close()


def test_joint_decoding_many_chunks_in_lockstep():
    """Ten chunks (the two golden chunks five times over) in one batch: the host halves of a frame run on the engine's thread
    pool (>= 8 chunks), the decoder steps carry rows of every chunk -- each copy must still give the golden answer."""
    case = JointCase("joint_tiny")
    run = case.js["runs"][2]
    x, lens = np.tile(case.x, (5, 1, 1)), np.tile(case.lens, 5)
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=16, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(x, lens, run["beam"], topk=int(run["pre_beam_ratio"] * run["beam"]))
    got = eng.joint_decode(run["ctc_weight"], run["length_bonus"], run["pre_beam_ratio"])
    assert len(got) == 10
    for b, g in enumerate(got):
        want = run["chunks"][b % 2]
        assert list(g.tokens) == want["tokens"] and list(g.times) == want["times"], b
        assert abs(g.score - want["score"]) <= 2e-2 + 1e-3 * abs(want["score"])
    eng.close()


def test_joint_decoding_needs_enough_topk():
    from reverb_amd._lib import RvbError
    case = JointCase("joint_tiny")
    eng = Engine(case.cfg, case.sd, dtype="f32", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(case.x, case.lens, 4)
    with pytest.raises(RvbError, match="pre-beam"):
        eng.joint_decode(0.3, 0.5)
    eng.close()


def test_joint_decoding_retries_with_more_logprobs_on_a_long_run_of_exact_ties():
    """a run of more than 8 log-probs tying EXACTLY with the pre-beam threshold used to fail the whole decode
    (RVB_E_UNSUPPORTED) after all the work was done.  A model whose CTC head gives 25 tokens the same logit in every frame
    (zero weight rows, equal biases: bit-equal in the reference and in the f32 engine) puts such a run at the threshold of every
    frame; the engine now encodes the batch again keeping 64 log-probs per frame and must return what the oracle -- which, like
    the reference, compares the whole row with the threshold (beam_search_timesync.py:268-270) -- returns."""
    import torch
    from oracle import model_ref as M, search_ref as S
    case = JointCase("joint_tiny")
    sd = dict(case.sd)
    w, b = np.array(sd["ctc.ctc_lo.weight"], np.float32), np.array(sd["ctc.ctc_lo.bias"], np.float32)
    w[20:45] = 0.0
    b[20:45] = 20.0                       # above every other token of this model: the top 25 of every frame tie exactly
    sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"] = w, b
    beam, ctc_weight, bonus = 4, 0.5, 0.5
    tsd, cat = M.to_torch_sd(sd), torch.tensor(case.cat)
    with torch.no_grad():
        enc, mask = M.encoder_forward(tsd, case.cfg, torch.from_numpy(case.x), torch.from_numpy(case.lens), cat)
        lp = M.ctc_logprobs(tsd, enc)
    lens = mask.squeeze(1).sum(1)
    assert int((lp[0, 0] == lp[0, 0, 20]).sum()) >= 25                     # the run of exact ties is really there
    want = S.joint_decoding(tsd, case.cfg, enc, lens, lp, ctc_weight, beam, 1.5, bonus, cat)
    eng = Engine(case.cfg, sd, dtype="f32", device=0, max_chunks=4, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.encode(case.x, case.lens, beam, topk=int(1.5 * beam) + 8)           # 14 per frame: 8 spare for ties, the run has 25
    got = eng.joint_decode(ctc_weight, bonus, 1.5)
    assert eng.topk == 64                                                   # the retry happened
    for g, wnt in zip(got, want):
        assert list(g.tokens) == list(wnt.tokens) and list(g.times) == list(wnt.times)
        assert abs(g.score - wnt.score) <= 2e-2 + 1e-3 * abs(wnt.score)
    eng.close()
