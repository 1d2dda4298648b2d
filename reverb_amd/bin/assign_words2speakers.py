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
    scan is seconds of Python): overlap tests and distances as numpy blocks, the three cases resolved exactly as above
    (tests/test_diarization_host.py checks equality with the per-word function)."""
    import numpy as np
    starts = np.asarray(starts, np.float64)
    ends = starts + np.asarray(durs, np.float64)
    n = len(starts)
    if not turns:
        return [""] * n
    S = np.array([t[0] for t in turns]); E = np.array([t[1] for t in turns])
    labels = [t[2] for t in turns]
    out: List[str] = [""] * n
    for b0 in range(0, n, block):
        st, en = starts[b0:b0 + block, None], ends[b0:b0 + block, None]
        touch = (S[None, :] < en) & (E[None, :] > st)
        hit = touch & (st < en)                       # a zero-length word overlaps nothing (the tree query is empty)
        cnt = hit.sum(1)
        first = hit.argmax(1)
        # no overlapping turn: the nearest one (first of equally near turns), intervaltree's distance_to
        dist = np.where(touch, 0.0, np.where(st < S[None, :], S[None, :] - en, st - E[None, :]))
        near = dist.argmin(1)
        for i in range(len(cnt)):
            if cnt[i] == 1:
                out[b0 + i] = labels[first[i]]
            elif cnt[i] == 0:
                out[b0 + i] = labels[near[i]]
            else:
                overlap = defaultdict(float)
                for j in np.nonzero(hit[i])[0]:
                    overlap[labels[j]] += min(en[i, 0], E[j]) - max(st[i, 0], S[j])
                out[b0 + i] = max(overlap, key=overlap.get)
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
