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
    with open(args.output_stm_transcription, 'w') as f:
        for _, channel, start, dur, token, _ in read_ctm(args.ctm_transcription):
            start, dur = float(start), float(dur)
            f.write(f'{keys[0]} 1 {speaker_for_segment(start, dur, turns)} {start:.3f} {(start + dur):.3f} {token}\n')


if __name__ == '__main__':
    main()
