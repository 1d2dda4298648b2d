"""TEST INFRASTRUCTURE: golden vectors of the streaming encoder from the UNMODIFIED reference (through oracle/ref_shim.py).

  tiny_ln_streaming     `model.encoder.forward_chunk_by_chunk(xs, chunk, left, cat_embs)` (encoder.py:343-402; attention
                        caches, positional offsets) on the language-specific `tiny_ln` model -- the encoder-level API takes
                        cat_embs, the `simulate_streaming` seam of ASRModel does not (asr_model.py:301-306) -- for several
                        (decoding_chunk_size, num_decoding_left_chunks); stored: the encoder output (every 4th frame), the
                        final attention-cache length and the greedy tokens of its CTC posteriors.
  tiny_plain_streaming  a model WITHOUT language-specific layers (dataset_conf.pass_cat_emb false), where the reference's own
                        `model.decode(..., simulate_streaming=True, decoding_chunk_size=N)` runs end to end: greedy, prefix
                        beam and rescoring results per chunk.
Writes tests/golden/tiny_{ln,plain}_streaming.{json,npz}.       python -m oracle.gen_golden_streaming
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
from oracle.gen_golden import build_reference_model, calibrate_beta, chunk_feats   # noqa: E402

ref_shim.install()
import torch                                      # noqa: E402

SETTINGS = [(16, -1), (16, 2), (7, 1), (5, 0), (64, 1)]
PLAIN = dict(dims="tiny", norm="layer_norm", seed=5, seconds=24.0, chunk=1200, beam=6, ctc_weight=0.3, reverse_weight=0.0)
PLAIN_SETTINGS = [(16, -1), (8, 2)]


def plain_config():
    cfg = synth.make_config(PLAIN["dims"], PLAIN["norm"])
    cfg["dataset_conf"]["pass_cat_emb"] = False
    return cfg


def main():
    from wenet.transformer.search import ctc_greedy_search
    torch.set_num_threads(8)
    # ---- encoder-level API on the language-specific model
    with open(os.path.join(GOLDEN, "tiny_ln.json")) as f:
        js = json.load(f)
    case = js["case"]
    cfg = synth.make_config(case["dims"], case["norm"])
    pcm = synth.synth_audio(case["seconds"], seed=1234 + case["seed"])
    feats = fbank_ref.fbank(pcm)                                   # 2998 frames: one stream of 748 encoder frames
    sd = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, js["beta"])
    model, _ = build_reference_model(cfg, sd)
    cat = torch.tensor(case["cat"])
    xs = torch.from_numpy(feats).unsqueeze(0)
    out, arrays = {"case": case, "frames": int(feats.shape[0]), "runs": []}, {}
    for cs, left in SETTINGS:
        with torch.no_grad():
            ys, masks = model.encoder.forward_chunk_by_chunk(xs, cs, left, cat_embs=cat)
            probs = model.ctc_logprobs(ys)
            greedy = ctc_greedy_search(probs, masks.squeeze(1).sum(1), 0)
            # the cache the last forward_chunk returned: replay the loop to read its size (forward_chunk_by_chunk drops it)
            att, cnn, offset = torch.zeros((0, 0, 0, 0)), torch.zeros((0, 0, 0, 0)), 0
            window, stride = (cs - 1) * 4 + 7, 4 * cs
            for cur in range(0, xs.size(1) - 7 + 1, stride):
                y, att, cnn = model.encoder.forward_chunk(xs[:, cur:min(cur + window, xs.size(1))], offset, cs * left, att, cnn,
                                                           cat_embs=cat)
                offset += y.size(1)
        out["runs"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left, "out_frames": int(ys.shape[1]),
                            "final_cache_frames": int(att.shape[2]), "greedy": list(map(int, greedy[0].tokens))})
        arrays[f"ys_{cs}_{left}".replace("-", "m")] = ys[0, ::4].numpy().copy()
        print("lsl", cs, left, ys.shape, att.shape, len(greedy[0].tokens))
    with open(os.path.join(GOLDEN, "tiny_ln_streaming.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(GOLDEN, "tiny_ln_streaming.npz"), **arrays)

    # ---- ASRModel.decode(simulate_streaming=True) on a model without language-specific layers
    cfg = plain_config()
    pcm = synth.synth_audio(PLAIN["seconds"], seed=1234 + PLAIN["seed"])
    x, lens = chunk_feats(fbank_ref.fbank(pcm), PLAIN["chunk"])
    model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, PLAIN["seed"], synth.CTC_GAMMA, 0.0))
    beta = calibrate_beta(model, x, lens, None)
    model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, PLAIN["seed"], synth.CTC_GAMMA, beta))
    modes = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]
    out = {"case": PLAIN, "beta": beta, "gamma": synth.CTC_GAMMA, "lens": lens.tolist(), "runs": []}
    for cs, left in PLAIN_SETTINGS:
        rows = []
        for b in range(len(lens)):          # the reference's streaming path asserts batch 1
            with torch.no_grad():
                res = model.decode(modes, torch.from_numpy(x[b:b + 1]), torch.from_numpy(lens[b:b + 1]), PLAIN["beam"],
                                   decoding_chunk_size=cs, num_decoding_left_chunks=left, ctc_weight=PLAIN["ctc_weight"],
                                   simulate_streaming=True, reverse_weight=PLAIN["reverse_weight"], blank_id=0,
                                   infos={"tasks": ["transcribe"], "langs": ["en"]})
            g, p, r = (res[m][0] for m in modes)
            rows.append({"greedy": list(map(int, g.tokens)), "prefix": list(map(int, p.tokens)), "prefix_times": list(map(int, p.times)),
                         "nbest": [list(map(int, h)) for h in p.nbest], "rescoring": list(map(int, r.tokens)),
                         "rescoring_score": float(r.score), "rescoring_times": list(map(int, r.times))})
        out["runs"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left, "chunks": rows})
        print("plain", cs, left, [len(c["greedy"]) for c in rows], [len(c["rescoring"]) for c in rows])
    with open(os.path.join(GOLDEN, "tiny_plain_streaming.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
