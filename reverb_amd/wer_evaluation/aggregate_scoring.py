#!/usr/bin/env python
"""Suite-level WER from a directory of per-file alignment logs.

Contract (what `asr/wer_evaluation/aggregate_scoring.py` of the reference does for its user, taken from its command
line and stdout, tests/golden/wer_evaluation.json):

    aggregate_scoring.py <dir>      reads every `<dir>/*.json`, takes its `wer.bestWER` block
                                    {numErrors, insertions, deletions, substitutions, numWordsInReference}
    stdout, four lines:             "<title>:\\t<count>/<reference words> = <percent with 2 decimals>%"
                                    for TOTAL WER, Insertion Rate, Deletion Rate, Substitution Rate
    no reference words at all:      RuntimeError

The substitution count of a file is derived as numErrors - insertions - deletions (the reference's convention; the
`substitutions` key only enters the count of correct words).  Written as a column sum over a small table rather than
a stateful aggregator: the whole suite is at most a few thousand rows.
"""
from __future__ import annotations

import json
import sys
from argparse import ArgumentParser
from pathlib import Path
from typing import Iterable, List, Mapping, NamedTuple

TITLES = ("TOTAL WER", "Insertion Rate", "Deletion Rate", "Substitution Rate")


class Totals(NamedTuple):
    """Word counts over the suite: errors = ins + dele + sub; `ref` is the denominator of every rate."""
    errors: int
    ins: int
    dele: int
    sub: int
    correct: int
    ref: int

    def rates(self) -> List[float]:
        if self.ref == 0:
            raise RuntimeError("Something went wrong! Cannot compute a rate when `reference_count` is 0.")
        return [n / self.ref for n in (self.errors, self.ins, self.dele, self.sub)]

    def summary(self) -> str:
        counts = (self.errors, self.ins, self.dele, self.sub)
        return "\n".join(f"{t}:\t{n}/{self.ref} = {r:3.2%}" for t, n, r in zip(TITLES, counts, self.rates()))

    def wer(self) -> float:
        return self.rates()[0]


def total(blocks: Iterable[Mapping[str, float]]) -> Totals:
    """Column sums of the `bestWER` blocks."""
    ins = dele = sub = correct = ref = 0
    for b in blocks:
        n_ref, n_ins, n_del = b["numWordsInReference"], b["insertions"], b["deletions"]
        ins, dele, ref = ins + n_ins, dele + n_del, ref + n_ref
        sub += b["numErrors"] - n_ins - n_del
        correct += n_ref - b["substitutions"] - n_del
    return Totals(ins + dele + sub, ins, dele, sub, correct, ref)


def read_blocks(directory: Path):
    for path in directory.glob("*.json"):
        with path.open("r") as f:
            yield json.load(f)["wer"]["bestWER"]


def aggregate(directory: Path) -> Totals:
    return total(read_blocks(Path(directory)))


def main(argv=None):
    p = ArgumentParser(description="Aggregate WER over a directory of alignment JSON logs (fstalign --json-log, or the "
                                   "builtin aligner of scoring_commands).")
    p.add_argument("fstalign_out", type=Path, help="directory holding one <name>.json alignment log per file")
    print(aggregate(p.parse_args(argv).fstalign_out).summary())


if __name__ == "__main__":
    main(sys.argv[1:])
