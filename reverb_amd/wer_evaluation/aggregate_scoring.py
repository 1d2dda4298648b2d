#!/usr/bin/env python
"""Same command line and output as the reference's asr/wer_evaluation/aggregate_scoring.py (:12-114): sums the
`wer.bestWER` blocks of every `*.json` in a directory and prints the four rates."""
from __future__ import annotations

import json
from argparse import ArgumentParser
from dataclasses import dataclass
from pathlib import Path
from typing import Dict


@dataclass
class WERAggregator:
    insertion_count: int = 0
    deletion_count: int = 0
    substitution_count: int = 0
    correct_count: int = 0
    reference_count: int = 0

    def update(self, a: Dict[str, float]):
        """One file's alignment statistics (fstalign's bestWER keys; aggregate_scoring.py:37-44)."""
        self.insertion_count += a["insertions"]
        self.deletion_count += a["deletions"]
        self.substitution_count += a["numErrors"] - a["insertions"] - a["deletions"]
        self.correct_count += a["numWordsInReference"] - a["substitutions"] - a["deletions"]
        self.reference_count += a["numWordsInReference"]

    @property
    def num_errors(self):
        return self.insertion_count + self.deletion_count + self.substitution_count

    def check_state(self):
        if self.reference_count == 0:
            raise RuntimeError("Something went wrong! Cannot compute a rate when `reference_count` is 0.")

    def _rate(self, count) -> float:
        self.check_state()
        return count / self.reference_count

    def insertion_rate(self) -> float:
        return self._rate(self.insertion_count)

    def deletion_rate(self) -> float:
        return self._rate(self.deletion_count)

    def substitution_rate(self) -> float:
        return self._rate(self.substitution_count)

    def wer(self) -> float:
        return self._rate(self.num_errors)

    def summary(self) -> str:
        def line(title, numerator, rate):
            return f"{title}:\t{numerator}/{self.reference_count} = {rate:3.2%}"
        return "\n".join([line("TOTAL WER", self.num_errors, self.wer()),
                          line("Insertion Rate", self.insertion_count, self.insertion_rate()),
                          line("Deletion Rate", self.deletion_count, self.deletion_rate()),
                          line("Substitution Rate", self.substitution_count, self.substitution_rate())])


def aggregate(directory: Path) -> WERAggregator:
    agg = WERAggregator()
    for path in directory.glob("*.json"):
        with path.open("r") as f:
            agg.update(json.load(f)["wer"]["bestWER"])
    return agg


def main(argv=None):
    p = ArgumentParser(description="Takes in directory of fstalign outputs and calculates the aggregate WER metric over the "
                                   "full test suite.")
    p.add_argument("fstalign_out", type=Path, help="Directory of alignment JSON logs (fstalign --json-log, or the builtin aligner).")
    print(aggregate(p.parse_args(argv).fstalign_out).summary())


if __name__ == "__main__":
    main()
