"""Native (C++) CTC prefix beam search vs the reference's own known answers and vs the oracle
restatement.  Runs on CPU: the search is host code behind the C ABI (rvb_test_prefix_beam)."""
import numpy as np
import torch

from reverb_amd import _lib
from reverb_amd._lib import fptr, iptr, dptr
from oracle import search_ref as S

# produced by running the unmodified reference
C1_LOGITS = [[2.0, 0.1, 0.0, -1.0, -1.0], [0.2, 2.5, 0.1, -1.0, -0.5], [0.3, 2.0, 0.2, -1.0, -0.5],
             [2.2, 0.0, 0.3, -0.5, -1.0], [0.1, 1.9, 0.2, -0.7, -1.0], [0.0, 0.2, 2.4, 0.1, -1.0],
             [1.5, 0.1, 1.4, 0.0, -1.0], [2.0, 0.0, 0.1, 0.2, 1.9]]


def native_prefix(lp: torch.Tensor, T: int, beam: int, blank: int = 0):
    lib = _lib.load()
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
    lib = _lib.load()
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
