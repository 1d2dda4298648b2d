#!/usr/bin/env python
"""Word -> speaker join of /root/reference/diarization/assign_words2speakers.py:25-89 (BASELINE config 5):
python -m reverb_amd.bin.assign_words2speakers diar.rttm words.ctm out.stm

The reference queries an `intervaltree.IntervalTree` (not installed here); the same three cases are
restated on a plain list of turns: exactly one overlapping turn -> its speaker; none -> the nearest
turn; several -> the speaker with the largest total overlap.  Where the reference's result depends on
set iteration order (exact ties) the earliest turn wins here.
"""
import argparse
import csv
from collections import defaultdict
from typing import List, Tuple

from reverb_amd.diarization import load_rttm

Turn = Tuple[float, float, str]


def read_ctm(ctm_path):
    with open(ctm_path, 'r') as f:
        for row in csv.reader(f, delimiter=' '):
            yield row


def make_turns(annotation) -> List[Turn]:
    turns = sorted({(seg.start, seg.end, label) for seg, _, label in annotation.itertracks(yield_label=True)})
    for s, e, _ in turns:
        if not s < e:
            raise ValueError(f"IntervalTree: Null Interval objects not allowed in IntervalTree: Interval({s}, {e})")
    return turns


def speaker_for_segment(start: float, dur: float, turns: List[Turn]) -> str:
    end = start + dur
    hits = [t for t in turns if t[0] < end and t[1] > start] if start < end else []
    if len(hits) == 1:
        return hits[0][2]
    if not hits:
        if not turns:
            return ""

        def distance(t):            # intervaltree.Interval.distance_to
            if t[0] < end and t[1] > start:
                return 0
            return t[0] - end if start < t[0] else start - t[1]
        return min(turns, key=distance)[2]
    overlap = defaultdict(float)
    for s, e, label in hits:
        overlap[label] += min(end, e) - max(start, s)
    return max(overlap, key=overlap.get)


def speakers_for_words(starts, durs, turns: List[Turn], block: int = 1024) -> List[str]:
    """`speaker_for_segment` for many words at once (an hour of speech is ~10^4 words against ~10^3 turns: the per-word
    scan is seconds of Python).  Round 5: candidate turns per word by two binary searches on the sorted turn list -- a
    turn overlaps [start, end) only if it starts before `end` (index < j1) and ends after `start` (the running maximum of
    the ends bounds the first such index, j0) -- so the work is the handful of (word, turn) pairs that can overlap, not
    words x turns (3 h of audio, 1 500 words x 6 200 turns: 100 ms -> a few ms).  The three cases are resolved exactly as
    in the per-word function (tests/test_diarization_host.py and tests/test_words2speakers_pin.py check equality); `turns`
    must be sorted (make_turns)."""
    import numpy as np
    starts = np.asarray(starts, np.float64)
    ends = starts + np.asarray(durs, np.float64)
    n = len(starts)
    if not turns:
        return [""] * n
    S = np.array([t[0] for t in turns]); E = np.array([t[1] for t in turns])
    labels = [t[2] for t in turns]
    out: List[str] = [""] * n
    if n == 0:
        return out
    if np.any(S[1:] < S[:-1]):                         # not sorted by start: the binary searches do not apply
        return [speaker_for_segment(float(s), float(e - s), turns) for s, e in zip(starts, ends)]
    j1 = np.searchsorted(S, ends, side="left")                              # turns [0, j1) start before the word ends
    j0 = np.minimum(np.searchsorted(np.maximum.accumulate(E), starts, side="right"), j1)    # turns [0, j0) end at or before its start
    lens = j1 - j0
    total = int(lens.sum())
    pw = np.repeat(np.arange(n), lens)
    pj = j0[pw] + (np.arange(total) - np.repeat(np.cumsum(lens) - lens, lens))
    ok = (E[pj] > starts[pw]) & (starts[pw] < ends[pw])                     # a zero-length word overlaps nothing (the tree query is empty)
    pw, pj = pw[ok], pj[ok]
    cnt = np.bincount(pw, minlength=n)
    first_at = np.searchsorted(pw, np.arange(n), side="left")              # pairs are ordered by (word, turn index)
    for i in np.nonzero(cnt == 1)[0]:
        out[i] = labels[pj[first_at[i]]]
    for i in np.nonzero(cnt > 1)[0]:                                        # overlapped speech: the speaker with the largest total overlap
        overlap = defaultdict(float)
        st, en = starts[i], ends[i]
        for j in pj[first_at[i]:first_at[i] + cnt[i]]:
            overlap[labels[j]] += min(en, E[j]) - max(st, S[j])
        out[i] = max(overlap, key=overlap.get)
    for i in np.nonzero(cnt == 0)[0]:                                       # no overlapping turn: the nearest one (first of equally near
        st, en = starts[i], ends[i]                                         # turns), intervaltree's distance_to over ALL turns
        touch = (S < en) & (E > st)
        dist = np.where(touch, 0.0, np.where(st < S, S - en, st - E))
        out[i] = labels[int(dist.argmin())]
    return out


def main(argv=None):
    parser = argparse.ArgumentParser('Assign words to speakers based on a diarization rttm file and ctm transcription')
    parser.add_argument('diarization_rttm', help='diarization rttm file')
    parser.add_argument('ctm_transcription', help='ctm transcription file')
    parser.add_argument('output_stm_transcription', help='output file in .stm format')
    args = parser.parse_args(argv)
    rttm = load_rttm(args.diarization_rttm)
    keys = list(rttm.keys())
    assert len(keys) == 1, keys
    turns = make_turns(rttm[keys[0]])
    rows = [(float(r[2]), float(r[3]), r[4]) for r in read_ctm(args.ctm_transcription)]
    who = speakers_for_words([r[0] for r in rows], [r[1] for r in rows], turns)
    with open(args.output_stm_transcription, 'w') as f:
        for (start, dur, token), spk in zip(rows, who):
            f.write(f'{keys[0]} 1 {spk} {start:.3f} {(start + dur):.3f} {token}\n')


if __name__ == '__main__':
    main()
