#!/usr/bin/env python
"""Same command line and output as the reference's asr/wer_evaluation/scoring_commands.py (:10-120): one alignment
command per hypothesis CTM on stdout.  `fstalign` may be the path of the fstalign binary, as in the reference, or the
word `builtin`, which selects `python -m reverb_amd.wer_evaluation.align` (same sub-command and flags)."""
from __future__ import annotations

import sys
from argparse import ArgumentParser
from pathlib import Path
from typing import Optional


def init_args(argv=None):
    p = ArgumentParser(description="Generates fstalign a list of commands that will perform adequate alignment between a "
                                   "test-suite and hypothesis directory. This script assumes the hypotheses are in CTM format "
                                   "and the references are in NLP format.")
    p.add_argument("fstalign", type=Path, help="Path to the fstalign binary, or `builtin` for reverb_amd's own aligner.")
    p.add_argument("ref", type=Path, help="Test suite transcript file or directory (NLP format).")
    p.add_argument("hyp", type=Path, help="ASR hypothesis file or directory (CTM format).")
    p.add_argument("out", type=Path, help="Output directory for the alignment JSON logs.")
    p.add_argument("--ref-norm", type=Path, default=None, help="Normalization file or directory (<name>.norm.json).")
    p.add_argument("--synonyms-file", type=Path, default=None, help="fstalign synonym file.")
    return p.parse_args(argv)


def prepare_IO(ref_path: Path, hyp_path: Path, out_path: Path, ref_norm_path: Optional[Path] = None):
    """(reference, hypothesis, json log, normalization) per hypothesis: a directory of CTMs pairs each `<name>.ctm` with
    `<ref>/<name>.nlp`, `<out>/<name>.log.json` and `<ref-norm>/<name>.norm.json`; a single file is taken as given."""
    out_path.mkdir(parents=True, exist_ok=True)
    if hyp_path.is_dir():
        for hyp_file in hyp_path.glob("**/*.ctm"):
            name = hyp_file.stem
            norm = (ref_norm_path / (name + ".norm.json")).resolve() if ref_norm_path else None
            yield (ref_path / (name + ".nlp")).resolve(), hyp_file.resolve(), (out_path / (name + ".log.json")).resolve(), norm
    else:
        yield (ref_path.resolve(), hyp_path.resolve(), (out_path / (hyp_path.stem + ".log.json")).resolve(),
               ref_norm_path.resolve() if ref_norm_path else None)


def commands(args):
    binary = [sys.executable, "-m", "reverb_amd.wer_evaluation.align"] if str(args.fstalign) == "builtin" else [str(args.fstalign)]
    for ref_file, hyp_file, out_file, norm_file in prepare_IO(args.ref, args.hyp, args.out, args.ref_norm):
        cmd = binary + ["wer", "--ref", str(ref_file), "--hyp", str(hyp_file), "--json-log", str(out_file)]
        if norm_file:
            cmd += ["--ref-json", str(norm_file)]
        if args.synonyms_file:
            cmd += ["--syn", str(args.synonyms_file)]
        yield " ".join(cmd)


def main(argv=None):
    for line in commands(init_args(argv)):
        print(line)


if __name__ == "__main__":
    main()
