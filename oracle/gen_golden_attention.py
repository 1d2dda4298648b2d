"""TEST INFRASTRUCTURE: golden vectors of the `attention` decoding mode (search.py:251-360), produced by the
UNMODIFIED reference through oracle/ref_shim.py for the cases of oracle/gen_golden.py (same weights: the
calibrated beta is read back from the case's json).  Writes tests/golden/<case>_attention.json.

    python -m oracle.gen_golden_attention        (needs /root/reference; not run on the GPU box)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import fbank_ref, ref_shim            # noqa: E402
from reverb_amd import synth                      # noqa: E402
from oracle.gen_golden import build_reference_model, chunk_feats   # noqa: E402  (puts the repo first on sys.path ...)

ref_shim.install()                                # ... so the reference's `wenet` is put in front afterwards
import torch                                      # noqa: E402

CASES = ["tiny_ln", "tiny_ln_r2l", "tiny_bn", "small_ln"]


def main():
    torch.set_num_threads(8)
    for name in CASES:
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            js = json.load(f)
        case = js["case"]
        cfg = synth.make_config(case["dims"], case["norm"])
        pcm = synth.synth_audio(case["seconds"], seed=1234 + case["seed"])
        x, lens = chunk_feats(fbank_ref.fbank(pcm), case["chunk"], case.get("tail_frames"))
        sd = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, js["beta"])
        model, _ = build_reference_model(cfg, sd)
        out = {"case": case, "beam": case["beam"], "runs": []}
        for lp in (0.0, 0.6):
            with torch.no_grad():
                res = model.decode(["attention"], torch.from_numpy(x), torch.from_numpy(lens), case["beam"], cat_embs=torch.tensor(case["cat"]),
                                   blank_id=0, length_penalty=lp, infos={"tasks": ["transcribe"], "langs": ["en"]})["attention"]
            out["runs"].append({"length_penalty": lp, "tokens": [list(map(int, r.tokens)) for r in res]})
            print(name, "lp", lp, "tokens/chunk", [len(r.tokens) for r in res])
        with open(os.path.join(GOLDEN, name + "_attention.json"), "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
