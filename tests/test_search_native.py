"""Native (C++) CTC prefix beam search vs the reference's own known answers and vs the oracle
restatement.  Runs on CPU: the search is host code behind the C ABI (rvb_test_prefix_beam)."""
import numpy as np
import pytest
import torch

from reverb_amd import _lib
from reverb_amd._lib import fptr, iptr, dptr
from oracle import search_ref as S

# produced by running the unmodified reference
C1_LOGITS = [[2.0, 0.1, 0.0, -1.0, -1.0], [0.2, 2.5, 0.1, -1.0, -0.5], [0.3, 2.0, 0.2, -1.0, -0.5],
             [2.2, 0.0, 0.3, -0.5, -1.0], [0.1, 1.9, 0.2, -0.7, -1.0], [0.0, 0.2, 2.4, 0.1, -1.0],
             [1.5, 0.1, 1.4, 0.0, -1.0], [2.0, 0.0, 0.1, 0.2, 1.9]]


def native_prefix(lp: torch.Tensor, T: int, beam: int, blank: int = 0):
    lib = _lib.load_test()
    tv, ti = lp.topk(beam, dim=-1)
    tv = np.ascontiguousarray(tv.numpy(), np.float32); ti = np.ascontiguousarray(ti.numpy(), np.int32)
    ml = max(T, 1)
    n = np.zeros(1, np.int32); toks = np.full((beam, ml), -1, np.int32); lens = np.zeros(beam, np.int32)
    times = np.full((beam, ml), -1, np.int32); tl = np.zeros(beam, np.int32); sc = np.zeros(beam, np.float64)
    _lib.check(lib.rvb_test_prefix_beam(fptr(tv), iptr(ti), T, beam, blank, iptr(n), iptr(toks), iptr(lens),
                                        iptr(times), iptr(tl), dptr(sc)))
    k = int(n[0])
    return ([tuple(toks[i, :lens[i]].tolist()) for i in range(k)], sc[:k].tolist(),
            [times[i, :tl[i]].tolist() for i in range(k)])


def test_kat_c1_reference_known_answer():
    lp = torch.tensor(C1_LOGITS).log_softmax(-1)
    nbest, scores, times = native_prefix(lp, 8, 3)
    assert nbest == [(1, 1, 2), (1, 1, 2, 4), (1, 2, 1, 2)]
    np.testing.assert_allclose(scores, [-2.75166, -2.85166, -3.916388], atol=2e-6)
    assert times == [[1, 4, 5], [1, 4, 5, 7], [1, 2, 4, 5]]
    # and the oracle restatement gives the same known answer
    o = S.ctc_prefix_beam_search(lp.unsqueeze(0), torch.tensor([8]), 3)[0]
    assert [tuple(x) for x in o.nbest] == nbest and o.nbest_times == times
    assert S.ctc_greedy_search(lp.unsqueeze(0), torch.tensor([8]))[0].tokens == [1, 1, 2]


def test_random_vs_oracle_bit_exact():
    g = torch.Generator().manual_seed(0)
    for trial in range(40):
        T, V, beam = 60, 50, [5, 10, 3][trial % 3]
        logits = torch.randn(T, V, generator=g) * 2
        logits[:, 0] += 2.5 + (trial % 4)          # blank-dominant like real CTC posteriors
        if trial % 5 == 0:
            logits[:, 0] -= 6                      # blank-poor: exercises the `vs_ns` quirk (times shorter than tokens)
        lp = logits.log_softmax(-1)
        n_t = T if trial % 7 else 0 if trial == 0 else T - 13
        o = S.ctc_prefix_beam_search(lp.unsqueeze(0), torch.tensor([n_t]), beam)[0]
        nbest, scores, times = native_prefix(lp, n_t, beam)
        assert nbest == [tuple(x) for x in o.nbest], trial
        assert times == o.nbest_times, trial
        assert scores == o.nbest_scores, trial      # float64 bit-exact (same libm, same order)


def test_empty_utterance():
    lp = torch.randn(4, 9).log_softmax(-1)
    nbest, scores, times = native_prefix(lp, 0, 3)
    assert nbest == [()] and scores == [0.0] and times == [[]]


def test_property_random_shapes_ties_and_tiny_beams():
    """The native search keys a frame's candidates by (beam entry, top-k token) instead of by token tuple and only
    creates trie nodes for survivors; prefixes that leave the beam and come back, exact score ties (quantised
    logits: the stable sort order decides) and beams wider than the vocabulary minus one are where that could
    differ from the reference's dictionary walk."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None, derandomize=True)
    @given(st.integers(0, 2 ** 31 - 1), st.integers(1, 40), st.integers(2, 12), st.integers(1, 6), st.sampled_from([0.0, 0.5, 1.0]),
           st.floats(-3.0, 6.0))
    def check(seed, T, V, beam, quant, blank_bias):
        beam = min(beam, V)
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(T, V, generator=g) * 2
        logits[:, 0] += blank_bias
        if quant:
            logits = (logits / quant).round() * quant           # many exactly equal log-probs
        lp = logits.log_softmax(-1)
        n_t = T - (seed % 3 if T > 3 else 0)
        o = S.ctc_prefix_beam_search(lp.unsqueeze(0), torch.tensor([n_t]), beam)[0]
        nbest, scores, times = native_prefix(lp, n_t, beam)
        assert nbest == [tuple(x) for x in o.nbest]
        assert times == o.nbest_times
        assert scores == o.nbest_scores

    check()


def test_native_search_on_the_reference_goldens():
    """The native search fed with the reference's own per-frame top-k log-probs (tests/golden/*.npz) must give the
    reference's n-best lists, float64 scores and peak frames of ctc_prefix_beam_search (tests/golden/*.json, both
    written by the unmodified reference in oracle/gen_golden.py) -- no oracle in between."""
    import pytest
    from tests.golden_util import CASES, Case
    lib = _lib.load_test()
    checked = 0
    for name in CASES:
        case = Case(name)
        tv_all, ti_all = case.arrays["topk_val"], case.arrays["topk_idx"]
        beam = case.beam
        for b, g in enumerate(case.golden("ctc_prefix_beam_search")):
            T = int(case.js["encoder_lens"][b])
            tv = np.ascontiguousarray(tv_all[b], np.float32); ti = np.ascontiguousarray(ti_all[b], np.int32)
            ml = max(T, 1)
            n = np.zeros(1, np.int32); toks = np.full((beam, ml), -1, np.int32); lens = np.zeros(beam, np.int32)
            times = np.full((beam, ml), -1, np.int32); tl = np.zeros(beam, np.int32); sc = np.zeros(beam, np.float64)
            _lib.check(lib.rvb_test_prefix_beam(fptr(tv), iptr(ti), T, beam, 0, iptr(n), iptr(toks), iptr(lens), iptr(times),
                                                iptr(tl), dptr(sc)))
            k = int(n[0])
            assert [toks[i, :lens[i]].tolist() for i in range(k)] == g["nbest"], (name, b)
            assert [times[i, :tl[i]].tolist() for i in range(k)] == g["nbest_times"], (name, b)
            assert sc[:k].tolist() == pytest.approx(g["nbest_scores"], rel=0, abs=1e-9), (name, b)
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("reversed_", [0, 1])
def test_rescoring_trie_against_a_python_trie(reversed_):
    """engine.hip build_trie (host code): one decoder row per distinct prefix of a chunk's hypotheses; every (hypothesis, j)
    pair must map to the row of ITS prefix and to its own target, a hypothesis' new rows must be a contiguous suffix of its
    path, and nothing may be shared across chunks."""
    import ctypes as C
    lib = _lib.load_test()
    rng = np.random.default_rng(5 + reversed_)
    sos = eos = 99
    hyps, chunk_of = [], []
    for chunk in range(4):
        base = rng.integers(1, 6, int(rng.integers(0, 30))).tolist()
        n = int(rng.integers(0, 7)) if chunk != 2 else 0            # chunk 2 has no hypothesis at all
        for i in range(n):
            h = list(base)
            for _ in range(int(rng.integers(0, 4))):                # a few edits: shared prefixes of every length, duplicates, prefixes
                k = int(rng.integers(0, len(h) + 1))
                if h and rng.random() < 0.5:
                    h = h[:k]
                else:
                    h = h[:k] + [int(rng.integers(1, 6))] + h[k:]
            hyps.append(h); chunk_of.append(chunk)
    hyps.append([]); chunk_of.append(3)                              # an empty hypothesis
    lens = np.array([len(h) for h in hyps], np.int32)
    toks = np.array([t for h in hyps for t in h] + [0], np.int32)
    P = int(lens.sum() + len(hyps))
    out = {k: np.full(n, -7, np.int32) for k, n in dict(tok=P, pos=P, path=P, hq_start=len(hyps), hq_len=len(hyps), hq_pos0=len(hyps),
                                                         tgt_ptr=P + 1, tgt=P, pair_slot=P).items()}
    n_rows, n_work = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.rvb_test_build_trie(iptr(toks), iptr(lens), iptr(np.array(chunk_of, np.int32)), len(hyps), 4, sos, eos, reversed_,
                                       C.byref(n_rows), *(iptr(out[k]) for k in ("tok", "pos", "path", "hq_start", "hq_len", "hq_pos0",
                                                                                 "tgt_ptr", "tgt", "pair_slot")), C.byref(n_work)))
    R = n_rows.value
    seqs = [h[::-1] if reversed_ else h for h in hyps]
    want_rows = len({(c, tuple(s[:j])) for s, c in zip(seqs, chunk_of) for j in range(len(s) + 1)})
    assert R == want_rows
    row_of, p = {}, 0
    for i, (s, c) in enumerate(zip(seqs, chunk_of)):
        path = out["path"][p:p + len(s) + 1].tolist()
        new = [r for r in path if r >= out["hq_start"][i]] if out["hq_len"][i] else []
        for j, r in enumerate(path):
            key = (c, tuple(s[:j]))
            assert row_of.setdefault(key, r) == r and 0 <= r < R                      # same prefix <-> same row
            assert out["tok"][r] == (sos if j == 0 else s[j - 1]) and out["pos"][r] == j
            slot = out["pair_slot"][p + j]
            assert out["tgt_ptr"][r] <= slot < out["tgt_ptr"][r + 1] and out["tgt"][slot] == (s[j] if j < len(s) else eos)
        # the rows this hypothesis adds: contiguous, a suffix of its path, starting at position hq_pos0
        n_own = int(out["hq_len"][i])
        assert path[len(path) - n_own:] == list(range(out["hq_start"][i], out["hq_start"][i] + n_own)) if n_own else True
        assert n_own == 0 or out["hq_pos0"][i] == len(path) - n_own
        p += len(s) + 1
    assert len(set(row_of.values())) == R and out["tgt_ptr"][R] == P and sorted(out["pair_slot"].tolist()) == list(range(P))
    assert n_work.value == sum(-(-int(n) // 16) for n in out["hq_len"])


# ------------------------------------------------------------------------------------------------ joint_decoding
def native_joint(sd, cfg, mem, lpz, run, cat, K=None):
    """Drive the native JointSearch state machine (search.cpp) for one chunk; the attention log-probs it asks for come from the
    oracle's decoder (model_ref.decoder_step) -- on the GPU the engine's batched decoder step takes that place."""
    import ctypes as C
    import torch.nn.functional as F
    from oracle import model_ref as M
    lib = _lib.load_test()
    V = cfg["output_dim"]
    sos, beam = V - 1, run["beam"]
    pre_beam = int(run["pre_beam_ratio"] * beam)
    K = K or pre_beam
    h = C.c_void_p(lib.rvb_test_joint_new(beam, pre_beam, 0, sos, run["ctc_weight"], 1.0 - run["ctc_weight"], run["length_bonus"]))
    n = mem.shape[1]
    mask = torch.ones(1, 1, n, dtype=torch.bool)
    rows, caches = {}, {}

    def decode(node):
        toks = np.zeros(4096, np.int32); ln = np.zeros(1, np.int32)
        _lib.check(lib.rvb_test_joint_prefix(h, node, iptr(toks), iptr(ln)))
        prefix = toks[:ln[0]].tolist()
        parent_cache = None if len(prefix) == 1 else caches[tuple(prefix[:-1])]
        out, cache = M.decoder_step(sd, cfg, "left_decoder", mem, mask, torch.tensor([prefix[-1]]), len(prefix) - 1, parent_cache, cat)
        caches[tuple(prefix)] = cache
        rows[node] = F.log_softmax(out, dim=-1)[0].numpy()

    with torch.no_grad():
        decode(0)
        cap = 1024
        dec = np.zeros(cap, np.int32); pn = np.zeros(cap, np.int32); pt = np.zeros(cap, np.int32)
        nd = np.zeros(1, np.int32); npairs = np.zeros(1, np.int32)
        tv_all, ti_all = lpz.topk(K, dim=-1)
        for t in range(n):
            tv = np.ascontiguousarray(tv_all[t].numpy(), np.float32); ti = np.ascontiguousarray(ti_all[t].numpy(), np.int32)
            rc = lib.rvb_test_joint_begin(h, t, fptr(tv), iptr(ti), K, float(lpz[t, 0]), float(lpz[t, 0]), iptr(dec), iptr(nd), iptr(pn),
                                          iptr(pt), iptr(npairs), cap)
            assert rc >= 0, lib.rvb_last_error()
            if rc == 0:
                continue
            for node in dec[:nd[0]]:
                decode(int(node))
            vals = np.array([rows[int(a)][int(b)] for a, b in zip(pn[:npairs[0]], pt[:npairs[0]])], np.float32)
            _lib.check(lib.rvb_test_joint_finish(h, fptr(vals)))
    toks = np.zeros(4096, np.int32); st = np.zeros(4096, np.int32); en = np.zeros(4096, np.int32); conf = np.zeros(4096, np.float64)
    ln = np.zeros(1, np.int32); score = np.zeros(1, np.float64)
    _lib.check(lib.rvb_test_joint_result(h, iptr(toks), iptr(st), iptr(en), dptr(conf), iptr(ln), dptr(score)))
    lib.rvb_test_joint_free(h)
    k = int(ln[0])
    return toks[:k].tolist(), st[:k].tolist(), en[:k].tolist(), conf[:k].tolist(), float(score[0])


@pytest.mark.parametrize("name", ["joint_tiny", "joint_small"])
def test_joint_decoding_native_state_machine_matches_reference_class(name):
    """search.cpp's JointSearch against the goldens of the reference's own BeamSearchTimeSync class (oracle/gen_golden_joint.py),
    with the oracle's decoder supplying the attention log-probs: tokens, start / end frames, confidences, joint score."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_util import JointCase
    from oracle import model_ref as M
    case = JointCase(name)
    sd = M.to_torch_sd(case.sd)
    cat = torch.tensor(case.cat)
    with torch.no_grad():
        enc, mask = M.encoder_forward(sd, case.cfg, torch.from_numpy(case.x), torch.from_numpy(case.lens), cat)
        probs = M.ctc_logprobs(sd, enc)
    lens = mask.squeeze(1).sum(1)
    for run in case.js["runs"]:
        bp = run.get("blank_penalty", 0.0)          # round 4: ctc_logprobs(encoder_out, blank_penalty, blank_id) feeds the mode
        lp = M.ctc_logprobs(sd, enc, bp, 0) if bp else probs
        pre_beam = int(run["pre_beam_ratio"] * run["beam"])
        for b, want in enumerate(run["chunks"]):
            n = int(lens[b])
            for K in (None, pre_beam + 8):  # the engine hands over the pre-beam + 8 more (ties with the threshold); same candidates
                toks, st, en, conf, score = native_joint(sd, case.cfg, enc[b:b + 1, :n], lp[b, :n], run, cat, K)
                assert toks == want["tokens"], (run, b, K)
                assert st == want["times"] and en == want["end_times"], (run, b, K)
                assert abs(score - want["score"]) < 2e-3 * max(1.0, abs(want["score"]))
                np.testing.assert_allclose(conf, want["tokens_confidence"], rtol=2e-3, atol=1e-6)


def test_host_pool_runs_every_item_once():
    """The worker pool behind the CTC search and the trie building (csrc/engine.h HostPool): jobs of varying width reuse the
    same threads; each work item of each job runs exactly once."""
    lib = _lib.load_test()
    for threads, items, rounds in ((1, 10, 3), (4, 1000, 30), (16, 37, 50), (32, 0, 5), (8, 100000, 4)):
        _lib.check(lib.rvb_test_host_pool(threads, items, rounds), "rvb_test_host_pool")
