"""The word -> speaker join pinned to the reference FUNCTION (VERDICT r4 "missing" #3).

`tests/golden/words2speakers.json` holds what the unmodified `/root/reference/diarization/assign_words2speakers.py:24-61`
returned (through `oracle/intervaltree_shim.py`, a stand-in for the two intervaltree methods it calls) on 2 372 seeded
(word, turn set) pairs covering its three branches; the product's per-word function and its vectorised form must give
the same speaker.  Where the reference's own answer depends on set iteration order -- several speakers with EXACTLY the
same overlap or several turns at EXACTLY the same distance -- any of the tied answers is accepted and the product's
documented rule (the earliest turn) is what it must return; such pairs are counted, not skipped.  With `/root/reference`
present the reference function is also executed live on fresh random pairs.
"""
import json
import os
import random
from collections import defaultdict

import pytest

from reverb_amd.bin import assign_words2speakers as A

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "words2speakers.json")


def _turn_list(turns):
    return sorted({(s, e, lab) for s, e, lab in turns})


def _tied_answers(start, dur, turns):
    """Every answer the reference can give, by brute force over the same float expressions."""
    end = start + dur
    hits = [t for t in turns if start < end and t[0] < end and t[1] > start]
    if len(hits) == 1:
        return {hits[0][2]}
    if not hits:
        if not turns:
            return {""}
        dist = [0 if (t[0] < end and t[1] > start) else (t[0] - end if start < t[0] else start - t[1]) for t in turns]
        m = min(dist)
        return {t[2] for t, d in zip(turns, dist) if d == m}
    tot = defaultdict(float)
    for s, e, lab in hits:
        tot[lab] += min(end, e) - max(start, s)
    m = max(tot.values())
    return {lab for lab, v in tot.items() if abs(v - m) <= 1e-12 * max(1.0, abs(m))}     # the sum order is the set's


def _check(turns, words, speakers):
    turns = _turn_list(turns)
    vec = A.speakers_for_words([w[0] for w in words], [w[1] for w in words], turns)
    ties = 0
    for (s, d), want, got_v in zip(words, speakers, vec):
        got = A.speaker_for_segment(s, d, turns)
        ok = _tied_answers(s, d, turns)
        assert want in ok, (s, d, want, ok)            # the brute force agrees with the reference
        if len(ok) == 1:
            assert got == want and got_v == want, (s, d, got, got_v, want)
        else:
            ties += 1
            assert got in ok and got_v == got, (s, d, got, got_v, ok)
    return ties


def test_join_matches_the_reference_function_on_the_golden_pairs():
    doc = json.load(open(GOLDEN))
    assert doc["pairs"] >= 1000 and min(doc["branches"].values()) >= 50, doc["branches"]
    ties = sum(_check(c["turns"], c["words"], c["speakers"]) for c in doc["cases"])
    # exact ties are common in overlapped speech (a word lying inside two speakers' turns gives both the word's whole
    # duration): 358 of the 2 372 pairs; the other 2 014 have one possible answer and must match it
    assert doc["pairs"] - ties >= 1000, ties


def test_join_matches_the_reference_function_live():
    from oracle import intervaltree_shim as shim
    if not shim.reference_available():
        pytest.skip("reference checkout not present (GPU box): the golden pairs above are the pin")
    from oracle.gen_golden_words2speakers import make_case
    # with the REAL intervaltree==3.1.0 when this machine holds it, else the stand-in classes
    ref = shim.load_reference_module()
    Tree, Iv, which = shim.tree_classes()
    rng = random.Random(7)
    for idx in range(3, 25):
        c = make_case(rng, idx)
        tree = Tree(Iv(s, e, lab) for s, e, lab in c["turns"])
        _check(c["turns"], c["words"], [ref.speaker_for_segment(s, d, tree) for s, d in c["words"]])


def test_golden_pairs_hold_under_the_real_intervaltree_package():
    """The golden speakers were generated through the stand-in; where the real package is on this disk, the unmodified
    reference function is run again over the SAME golden pairs on the real `IntervalTree`: every answer must be one of the
    tied-possible answers, and equal to the golden one wherever only one answer is possible.  Also: the stand-in and the
    real tree return the same sets for random queries (so the stand-in is a faithful scan of the same intervals)."""
    from oracle import intervaltree_shim as shim
    real = shim.real_intervaltree()
    if real is None or not shim.reference_available():
        pytest.skip("no intervaltree==3.1.0 on this machine (or no reference checkout)")
    ref = shim.load_reference_module(real=True)
    doc = json.load(open(GOLDEN))
    checked = 0
    for c in doc["cases"]:
        turns = _turn_list(c["turns"])
        tree = real.IntervalTree(real.Interval(s, e, lab) for s, e, lab in turns)
        for (s, d), want in zip(c["words"], c["speakers"]):
            got = ref.speaker_for_segment(s, d, tree)
            ok = _tied_answers(s, d, turns)
            assert got in ok and (len(ok) > 1 or got == want), (s, d, got, want, ok)
            checked += 1
    assert checked == doc["pairs"]
    rng = random.Random(3)
    for _ in range(50):
        ivs = []
        for _ in range(rng.randrange(1, 30)):
            a = round(rng.uniform(0, 50), 2)
            ivs.append((a, a + round(rng.uniform(0.01, 8), 2), rng.choice("ABCD")))
        rt = real.IntervalTree(real.Interval(*iv) for iv in ivs)
        st = shim.IntervalTree(shim.Interval(*iv) for iv in ivs)
        assert len(rt) == len(st)
        for _ in range(40):
            a = round(rng.uniform(-1, 60), 2)
            b = a + rng.choice([0.0, 0.01, 1.0, 7.5])
            assert {tuple(iv) for iv in rt[a:b]} == {tuple(iv) for iv in st[a:b]}
            q = real.Interval(a, b + 0.01)
            for iv in ivs:
                assert real.Interval(*iv).distance_to(q) == shim.Interval(*iv).distance_to(shim.Interval(a, b + 0.01))


def test_script_writes_the_stm_the_reference_script_writes(tmp_path):
    """The reference file run as `__main__`, unmodified (its CTM reader, its tree construction from the annotation, its STM
    line format), against the product's `main` on the same RTTM + CTM.  pyannote's `load_rttm` is stood in for by the
    product's own reader -- that part is shared, not pinned; turn sets without exact ties so both must agree line by line."""
    from oracle import intervaltree_shim as shim
    if not shim.reference_available():
        pytest.skip("reference checkout not present")
    from reverb_amd.diarization import load_rttm
    rng = random.Random(11)
    t, rttm, ctm = 0.0, [], []
    for i in range(40):
        t += rng.uniform(0.1, 3.0)
        d = rng.uniform(0.3, 5.0)
        rttm.append(f"SPEAKER rec1 1 {t:.3f} {d:.3f} <NA> <NA> SPEAKER_{rng.randrange(4):02d} <NA> <NA>")
        t += d * rng.uniform(0.5, 1.0)
    w = 0.0
    while w < t + 3.0:
        d = round(rng.uniform(0.013, 0.7), 2)
        ctm.append(f"rec1.wav 0 {w:.2f} {d:.2f} w{len(ctm)} {rng.random():.2f}")
        w += d + rng.choice([0.0, 0.0, 0.137, 1.731])
    (tmp_path / "a.rttm").write_text("\n".join(rttm) + "\n")
    (tmp_path / "a.ctm").write_text("\n".join(ctm) + "\n")
    shim.run_reference_script(str(tmp_path / "a.rttm"), str(tmp_path / "a.ctm"), str(tmp_path / "ref.stm"), load_rttm)
    A.main([str(tmp_path / "a.rttm"), str(tmp_path / "a.ctm"), str(tmp_path / "got.stm")])
    want, got = (tmp_path / "ref.stm").read_text(), (tmp_path / "got.stm").read_text()
    assert len(want.splitlines()) == len(ctm) > 100
    assert got == want


def test_shim_refuses_null_intervals_like_the_package():
    from oracle import intervaltree_shim as shim
    with pytest.raises(ValueError, match="Null Interval"):
        shim.IntervalTree([shim.Interval(1.0, 1.0, "A")])
    t = shim.IntervalTree([shim.Interval(0.0, 1.0, "A"), shim.Interval(0.0, 1.0, "A"), shim.Interval(1.0, 2.0, "B")])
    assert len(t) == 2 and {iv.data for iv in t[0.5:1.0]} == {"A"} and t[1.0:1.0] == set()
    assert shim.Interval(0.0, 1.0).distance_to(shim.Interval(1.0, 2.0)) == 0.0       # touching: gap of zero, not "overlap"
    assert shim.Interval(3.0, 4.0).distance_to(shim.Interval(1.0, 2.0)) == 1.0
