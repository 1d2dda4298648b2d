"""TEST INFRASTRUCTURE: golden vectors of chunk-masked decoding (`--decoding_chunk_size N --num_decoding_left_chunks M`,
recognize_wav.py:95-112 -> BaseEncoder.forward -> add_optional_chunk_mask) from the UNMODIFIED reference through
oracle/ref_shim.py, for the `tiny_ln` case with `encoder_conf.use_dynamic_chunk: true`.
Writes tests/golden/tiny_ln_chunkmask.{json,npz}.       python -m oracle.gen_golden_chunkmask
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
from oracle.gen_golden import build_reference_model, chunk_feats   # noqa: E402

ref_shim.install()
import torch                                      # noqa: E402

SETTINGS = [(16, -1), (16, 2), (7, 1), (-1, -1)]


def main():
    torch.set_num_threads(8)
    with open(os.path.join(GOLDEN, "tiny_ln.json")) as f:
        js = json.load(f)
    case = js["case"]
    cfg = synth.make_config(case["dims"], case["norm"])
    cfg["encoder_conf"]["use_dynamic_chunk"] = True
    pcm = synth.synth_audio(case["seconds"], seed=1234 + case["seed"])
    x, lens = chunk_feats(fbank_ref.fbank(pcm), case["chunk"], case.get("tail_frames"))
    sd = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, js["beta"])
    model, _ = build_reference_model(cfg, sd)
    cat = torch.tensor(case["cat"])
    out, arrays = {"case": case, "runs": []}, {}
    for cs, left in SETTINGS:
        with torch.no_grad():
            res = model.decode(["ctc_greedy_search", "attention_rescoring"], torch.from_numpy(x), torch.from_numpy(lens), case["beam"],
                               decoding_chunk_size=cs, num_decoding_left_chunks=left, ctc_weight=case["ctc_weight"], cat_embs=cat,
                               blank_id=0, infos={"tasks": ["transcribe"], "langs": ["en"]})
            enc, _ = model.encoder(torch.from_numpy(x), torch.from_numpy(lens), cs, left, cat_embs=cat)
        out["runs"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left,
                            "greedy": [list(map(int, r.tokens)) for r in res["ctc_greedy_search"]],
                            "rescoring": [list(map(int, r.tokens)) for r in res["attention_rescoring"]]})
        arrays[f"enc_{cs}_{left}".replace("-", "m")] = enc.numpy()[:, ::4].copy()
        print(cs, left, [len(r.tokens) for r in res["ctc_greedy_search"]])
    with open(os.path.join(GOLDEN, "tiny_ln_chunkmask.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(GOLDEN, "tiny_ln_chunkmask.npz"), **arrays)


if __name__ == "__main__":
    main()
