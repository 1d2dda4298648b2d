"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the word -> speaker join (BASELINE config 5).

Runs the UNMODIFIED `/root/reference/diarization/assign_words2speakers.py:24-61` (`speaker_for_segment`) through
`oracle/intervaltree_shim.py` on seeded random (word, turn set) pairs and writes
`tests/golden/words2speakers.json`: the turn sets, the words and the speaker the reference returned for each.

    python -m oracle.gen_golden_words2speakers

Turn sets: 0 .. 60 turns of 2 .. 6 speakers, some with heavy overlap, some sparse (gaps of seconds), times either
free floats or rounded to the millisecond grid an RTTM file carries (`{start:.3f} {dur:.3f}`), where exact ties in
overlap / distance do occur.  Words: CTM-like (start on a 10 ms grid or free, duration 0 .. 1.2 s, incl. zero-length
words and words outside every turn).  Every branch of the function is counted in the file's header.
"""
import json
import os
import random

from oracle import intervaltree_shim as shim

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "words2speakers.json")


def make_case(rng, idx):
    grid = idx % 2 == 1                          # odd cases live on the RTTM / CTM grids
    n_turns = [0, 1, 2][idx] if idx < 3 else rng.randint(3, 60)
    n_spk = rng.randint(2, 6)
    dense = rng.random() < 0.5
    t, turns = 0.0, []
    for _ in range(n_turns):
        t += rng.uniform(-2.0, 0.5) if dense else rng.uniform(-0.5, 4.0)
        t = max(t, 0.0)
        dur = rng.uniform(0.05, 6.0)
        s, e = (round(t, 3), round(t + dur, 3)) if grid else (t, t + dur)
        if not s < e:
            e = s + 0.001
        turns.append([s, e, "SPEAKER_%02d" % rng.randrange(n_spk)])
        t += dur * rng.uniform(0.2, 1.0)
    if n_turns > 4:
        turns.append(list(turns[2]))             # an exact duplicate collapses in the tree's set
        turns.append([turns[3][0], turns[3][1], "SPEAKER_%02d" % ((int(turns[3][2][-2:]) + 1) % n_spk)])   # same span, other speaker
    horizon = max([e for _, e, _ in turns], default=10.0) + 5.0
    words = []
    for _ in range(50):
        s = rng.uniform(-1.0, horizon)
        d = rng.choice([0.0, 0.01, 0.05]) if rng.random() < 0.1 else rng.uniform(0.0, 1.2)
        if grid:
            s, d = round(s, 2), round(d, 2)
        words.append([s, d])
    if turns:                                     # words pinned to turn boundaries (half-open interval semantics)
        a = turns[rng.randrange(len(turns))]
        words += [[a[1], 0.3], [a[0] - 0.3, 0.3], [a[0], 0.0], [a[0], a[1] - a[0]]]
    return {"grid": grid, "turns": turns, "words": words}


def main():
    ref = shim.load_reference_module()
    rng = random.Random(20240926)
    cases, branch = [], {"one": 0, "none": 0, "many": 0, "no_turns": 0}
    for idx in range(44):
        c = make_case(rng, idx)
        tree = shim.IntervalTree(shim.Interval(s, e, lab) for s, e, lab in c["turns"])
        c["speakers"] = []
        for s, d in c["words"]:
            k = len(tree[s:s + d])
            branch["no_turns" if not c["turns"] else "one" if k == 1 else "none" if k == 0 else "many"] += 1
            c["speakers"].append(ref.speaker_for_segment(s, d, tree))
        cases.append(c)
    doc = {"made_by": "oracle/gen_golden_words2speakers.py",
           "reference": "diarization/assign_words2speakers.py:24-61 (unmodified) through oracle/intervaltree_shim.py",
           "pairs": sum(len(c["words"]) for c in cases), "branches": branch, "cases": cases}
    with open(OUT, "w") as f:
        json.dump(doc, f)
    print("wrote", os.path.normpath(OUT), doc["pairs"], "pairs", branch)


if __name__ == "__main__":
    main()
