"""WER glue (reverb_amd/wer_evaluation) against the reference's own scripts (asr/wer_evaluation/*.py, pure Python): the same
inputs give the same stdout -- checked live when /root/reference is present, and against stdout captured from it
(tests/golden/wer_evaluation.json, written by this file's __main__) everywhere.  The built-in aligner is checked against a
brute-force alignment and a python-Levenshtein on random sequences."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from reverb_amd.wer_evaluation import aggregate_scoring as AG, align as AL, scoring_commands as SC

GOLD = os.path.join(ROOT, "tests", "golden", "wer_evaluation.json")
REF = "/root/reference/asr/wer_evaluation"
LOGS = [dict(insertions=3, deletions=5, substitutions=7, numErrors=15, numWordsInReference=120),
        dict(insertions=0, deletions=2, substitutions=1, numErrors=3, numWordsInReference=40),
        dict(insertions=11, deletions=0, substitutions=4, numErrors=15, numWordsInReference=77)]


def _make_tree(d):
    for sub in ("ref", "hyp", "norm", "logs"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    for name in ("a1", "b2"):
        open(os.path.join(d, "hyp", name + ".ctm"), "w").write(f"{name} 0 0.10 0.20 hello 1.00\n")
        open(os.path.join(d, "ref", name + ".nlp"), "w").write("token|speaker|ts|endTs|punctuation|case|tags|wer_tags\nhello|1|||||[]|[]\n")
    for i, l in enumerate(LOGS):
        json.dump({"wer": {"bestWER": l}}, open(os.path.join(d, "logs", f"f{i}.log.json"), "w"))


def _ours(d):
    import contextlib
    import io
    out = {}
    for key, argv in _cases(d).items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            (AG if key == "aggregate" else SC).main(argv[1:])
        out[key] = sorted(buf.getvalue().replace(d, "<D>").splitlines()) if key != "aggregate" else buf.getvalue().splitlines()
    return out


def _cases(d):
    return {"aggregate": ["aggregate_scoring.py", os.path.join(d, "logs")],
            "dir": ["scoring_commands.py", "bin/fstalign", os.path.join(d, "ref"), os.path.join(d, "hyp"), os.path.join(d, "out"),
                    "--ref-norm", os.path.join(d, "norm"), "--synonyms-file", "syn.txt"],
            "file": ["scoring_commands.py", "bin/fstalign", os.path.join(d, "ref", "a1.nlp"), os.path.join(d, "hyp", "a1.ctm"),
                     os.path.join(d, "out2")]}


def _reference(d):
    out = {}
    for key, argv in _cases(d).items():
        r = subprocess.run([sys.executable, os.path.join(REF, argv[0])] + argv[1:], capture_output=True, text=True, check=True)
        lines = r.stdout.replace(d, "<D>").splitlines()
        out[key] = sorted(lines) if key != "aggregate" else lines
    return out


def test_scripts_print_what_the_reference_prints(tmp_path):
    d = str(tmp_path)
    _make_tree(d)
    got = _ours(d)
    got["file"] = [l.replace(os.getcwd(), "<CWD>") for l in got["file"]]
    with open(GOLD) as f:
        want = json.load(f)
    assert got["aggregate"] == want["aggregate"] and got["dir"] == want["dir"]
    assert [l.split("syn.txt")[0] for l in got["file"]] == [l.split("syn.txt")[0] for l in want["file"]]
    assert got["aggregate"][0] == "TOTAL WER:\t33/237 = 13.92%"
    if os.path.isdir(REF):
        live = _reference(d)
        assert got["aggregate"] == live["aggregate"] and got["dir"] == live["dir"]


def _brute(ref, hyp):
    n, m = len(ref), len(hyp)
    dp = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(n + 1):
        dp[i][0] = i
    for j in range(m + 1):
        dp[0][j] = j
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            dp[i][j] = min(dp[i - 1][j - 1] + (ref[i - 1] != hyp[j - 1]), dp[i - 1][j] + 1, dp[i][j - 1] + 1)
    return dp[n][m]


def test_builtin_aligner_counts():
    rng = random.Random(3)
    for trial in range(60):
        n, m = rng.randint(0, 40), rng.randint(0, 40)
        ref = [rng.choice("abcdefg") for _ in range(n)]
        hyp = [rng.choice("abcdefg") for _ in range(m)]
        c = AL.wer_counts(ref, hyp)
        assert c["numErrors"] == _brute(ref, hyp) == c["substitutions"] + c["deletions"] + c["insertions"]
        assert n - c["deletions"] == m - c["insertions"]          # aligned pairs on both sides
    assert AL.wer_counts("the cat sat".split(), "The cat sat".split())["numErrors"] == 0      # case-insensitive
    assert AL.wer_counts("a b c d".split(), "a x c".split()) == dict(numErrors=2, substitutions=1, deletions=1, insertions=0)
    assert AL.wer_counts([], "a b".split()) == dict(numErrors=2, substitutions=0, deletions=0, insertions=2)


def test_builtin_pipeline_end_to_end(tmp_path):
    """scoring_commands builtin -> run the commands -> aggregate: NLP reference against a CTM hypothesis."""
    d = tmp_path
    (d / "ref").mkdir(); (d / "hyp").mkdir()
    (d / "ref" / "talk.nlp").write_text("token|speaker|ts|endTs|punctuation|case|tags|wer_tags\n" +
                                        "".join(f"{w}|1|||||[]|[]\n" for w in "we had a great quarter overall".split()))
    (d / "hyp" / "talk.ctm").write_text("".join(f"talk 0 {i}.00 0.50 {w} 0.90\n" for i, w in
                                                enumerate("we had great quarters over all".split())))
    args = SC.init_args(["builtin", str(d / "ref"), str(d / "hyp"), str(d / "out")])
    cmds = list(SC.commands(args))
    assert len(cmds) == 1 and " wer --ref " in cmds[0] and cmds[0].endswith("talk.log.json")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run(cmds[0].split(), check=True, env=env, cwd=ROOT)
    log = json.loads((d / "out" / "talk.log.json").read_text())["wer"]["bestWER"]
    assert log["numWordsInReference"] == 6 and log["numErrors"] == 4          # a deleted, quarter->quarters, overall->over, +all
    agg = AG.aggregate(d / "out")
    assert agg.summary().splitlines()[0] == "TOTAL WER:\t4/6 = 66.67%"


if __name__ == "__main__":      # regenerate the golden stdout from the reference's scripts
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        _make_tree(d)
        ref = _reference(d)
        ref["file"] = [l.replace(os.getcwd(), "<CWD>") for l in ref["file"]]
        json.dump(ref, open(GOLD, "w"), indent=1)
        print(json.dumps(ref, indent=1))
