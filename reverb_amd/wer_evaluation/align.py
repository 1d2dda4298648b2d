#!/usr/bin/env python
"""Built-in stand-in for `fstalign wer` (external binary the reference's scoring_commands.py drives; absent here):

    python -m reverb_amd.wer_evaluation.align wer --ref R --hyp H --json-log OUT [--ref-json N] [--syn S]

R: Rev NLP file (`token|speaker|ts|endTs|punctuation|case|tags|wer_tags` rows), CTM or plain text; H: CTM or plain text.
Words are compared case-insensitively; the minimal Levenshtein alignment is counted natively (librvb `rvb_wer_counts`,
host code).  What fstalign adds on top -- synonym rules (--syn) and reference normalisations (--ref-json) compiled into
FSTs -- is NOT reproduced: both flags are accepted and reported as ignored in the log.  The JSON log carries the
`wer.bestWER` block aggregate_scoring.py reads."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
from pathlib import Path
from typing import List

import numpy as np


def read_words(path: Path) -> List[str]:
    text = path.read_text(encoding="utf8")
    lines = [l for l in text.splitlines() if l.strip()]
    suffix = path.suffix.lower()
    if suffix == ".nlp" or (lines and lines[0].startswith("token|")):
        rows = lines[1:] if lines and lines[0].startswith("token|") else lines
        return [r.split("|", 1)[0] for r in rows if r.split("|", 1)[0]]
    if suffix == ".ctm":
        return [l.split()[4] for l in lines if len(l.split()) >= 5]
    return text.split()


def wer_counts(ref: List[str], hyp: List[str]):
    """{errors, substitutions, deletions, insertions} of hyp against ref (case-insensitive)."""
    from .. import _lib
    lib = _lib.load()
    ids = {}
    r = np.array([ids.setdefault(w.lower(), len(ids)) for w in ref], np.int32)
    h = np.array([ids.setdefault(w.lower(), len(ids)) for w in hyp], np.int32)
    out = (C.c_int64 * 4)()
    _lib.check(lib.rvb_wer_counts(_lib.iptr(r) if len(r) else None, len(r), _lib.iptr(h) if len(h) else None, len(h), out),
               "rvb_wer_counts")
    return dict(numErrors=int(out[0]), substitutions=int(out[1]), deletions=int(out[2]), insertions=int(out[3]))


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("command", choices=["wer"])
    p.add_argument("--ref", type=Path, required=True)
    p.add_argument("--hyp", type=Path, required=True)
    p.add_argument("--json-log", type=Path, default=None)
    p.add_argument("--ref-json", type=Path, default=None)
    p.add_argument("--syn", type=Path, default=None)
    a = p.parse_args(argv)
    ref, hyp = read_words(a.ref), read_words(a.hyp)
    best = wer_counts(ref, hyp)
    best["numWordsInReference"] = len(ref)
    best["numWordsInHypothesis"] = len(hyp)
    best["wer"] = best["numErrors"] / len(ref) if ref else 0.0
    log = {"wer": {"bestWER": best},
           "aligner": "reverb_amd.wer_evaluation.align (plain Levenshtein, case-insensitive)",
           "ignored": [str(x) for x in (a.ref_json, a.syn) if x is not None]}
    if a.json_log:
        a.json_log.parent.mkdir(parents=True, exist_ok=True)
        a.json_log.write_text(json.dumps(log, indent=1))
    print(f"WER: {best['numErrors']}/{len(ref)} = {best['wer']:.4f}  (ins {best['insertions']} del {best['deletions']} "
          f"sub {best['substitutions']})", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
