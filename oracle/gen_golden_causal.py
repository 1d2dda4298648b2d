"""TEST INFRASTRUCTURE: golden vectors of a CAUSAL-convolution model (`encoder_conf.causal: true`, `use_dynamic_chunk: true`
-- the stock WeNet U2++ streaming recipe) from the UNMODIFIED reference through oracle/ref_shim.py.  The causal
ConvolutionModule pads cnn_module_kernel-1 frames on the left only and hands a cnn_cache from chunk to chunk
(transformer/convolution.py:55-57,113-121; encoder.py:231-341; encoder_layer.py:164-244).

  tiny_causal        language-specific tiny model, LayerNorm conv module, K = 15:
                       offline   `model.decode` (greedy / prefix beam / rescoring) of a 2-chunk padded batch for several
                                 (decoding_chunk_size, num_decoding_left_chunks) + the encoder output
                       streaming `model.encoder.forward_chunk_by_chunk(xs, chunk, left, cat_embs)` for five settings, one
                                 with chunks SHORTER than the cnn cache (3 < 14 frames: the cache shifts), + the final
                                 att / cnn cache the reference returned
  tiny_causal_plain  no language-specific layers, BatchNorm conv module, EVEN kernel K = 8 (allowed when causal):
                       `model.decode(..., simulate_streaming=True, decoding_chunk_size=N)` end to end, and offline decode
Writes tests/golden/tiny_causal{,_plain}.{json,npz}.       python -m oracle.gen_golden_causal
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

MODES = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]
LSL = dict(name="tiny_causal", dims="tiny", norm="layer_norm", seed=7, seconds=30.0, chunk=2051, beam=8, ctc_weight=0.3,
           reverse_weight=0.3, cat=[0.6, 0.4], causal=True, use_dynamic_chunk=True, cnn_module_kernel=15)
LSL_OFFLINE = [(-1, -1), (16, 2), (8, -1)]
LSL_STREAM = [(16, -1), (16, 2), (7, 1), (3, 0), (64, 1)]
PLAIN = dict(name="tiny_causal_plain", dims="tiny", norm="batch_norm", seed=8, seconds=24.0, chunk=1200, beam=6, ctc_weight=0.3,
             reverse_weight=0.0, cat=[1.0, 0.0], causal=True, use_dynamic_chunk=True, cnn_module_kernel=8, pass_cat_emb=False)
PLAIN_STREAM = [(16, -1), (8, 2), (2, 3)]


def case_config(c):
    return synth.make_config(c["dims"], c["norm"], causal=c["causal"], use_dynamic_chunk=c["use_dynamic_chunk"],
                             cnn_module_kernel=c["cnn_module_kernel"], pass_cat_emb=c.get("pass_cat_emb", True))


def rows_of(res):
    out = []
    for b in range(len(res[MODES[0]])):
        g, p, r = (res[m][b] for m in MODES)
        out.append({"greedy": list(map(int, g.tokens)), "prefix": list(map(int, p.tokens)), "prefix_times": list(map(int, p.times)),
                    "nbest": [list(map(int, h)) for h in p.nbest], "nbest_scores": [float(v) for v in p.nbest_scores],
                    "rescoring": list(map(int, r.tokens)), "rescoring_score": float(r.score),
                    "rescoring_times": list(map(int, r.times))})
    return out


def calibrated_model(c, x, lens, cat):
    cfg = case_config(c)
    model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, c["seed"], synth.CTC_GAMMA, 0.0))
    beta = calibrate_beta(model, x, lens, cat)
    model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, c["seed"], synth.CTC_GAMMA, beta))
    assert model.encoder.encoders[0].conv_module.lorder == c["cnn_module_kernel"] - 1
    return model, beta


def main():
    from wenet.transformer.search import ctc_greedy_search
    torch.set_num_threads(8)
    infos = {"tasks": ["transcribe"], "langs": ["en"]}

    # ---------------------------------------------------------------- language-specific causal model
    c = LSL
    feats = fbank_ref.fbank(synth.synth_audio(c["seconds"], seed=1234 + c["seed"]))
    x, lens = chunk_feats(feats, c["chunk"])
    cat = torch.tensor(c["cat"])
    model, beta = calibrated_model(c, x, lens, cat)
    out = {"case": c, "beta": beta, "gamma": synth.CTC_GAMMA, "lens": lens.tolist(), "frames": int(feats.shape[0]),
           "offline": [], "streaming": []}
    arrays = {}
    for cs, left in LSL_OFFLINE:
        with torch.no_grad():
            res = model.decode(MODES, torch.from_numpy(x), torch.from_numpy(lens), c["beam"], decoding_chunk_size=cs,
                               num_decoding_left_chunks=left, ctc_weight=c["ctc_weight"], reverse_weight=c["reverse_weight"],
                               cat_embs=cat, blank_id=0, infos=infos)
            enc, mask = model.encoder(torch.from_numpy(x), torch.from_numpy(lens), cs, left, cat_embs=cat)
        out["offline"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left, "chunks": rows_of(res),
                               "encoder_lens": mask.squeeze(1).sum(1).tolist()})
        arrays[f"enc_{cs}_{left}".replace("-", "m")] = enc.numpy()[:, ::4].copy()
        print("offline", cs, left, [len(r["greedy"]) for r in out["offline"][-1]["chunks"]])
    xs = torch.from_numpy(feats).unsqueeze(0)
    for cs, left in LSL_STREAM:
        with torch.no_grad():
            ys, masks = model.encoder.forward_chunk_by_chunk(xs, cs, left, cat_embs=cat)
            probs = model.ctc_logprobs(ys)
            greedy = ctc_greedy_search(probs, masks.squeeze(1).sum(1), 0)
            # replay the loop to read the caches forward_chunk_by_chunk drops
            att, cnn, offset = torch.zeros((0, 0, 0, 0)), torch.zeros((0, 0, 0, 0)), 0
            window, stride = (cs - 1) * 4 + 7, 4 * cs
            for cur in range(0, xs.size(1) - 7 + 1, stride):
                y, att, cnn = model.encoder.forward_chunk(xs[:, cur:min(cur + window, xs.size(1))], offset, cs * left, att, cnn,
                                                           cat_embs=cat)
                offset += y.size(1)
        key = f"{cs}_{left}".replace("-", "m")
        out["streaming"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left, "out_frames": int(ys.shape[1]),
                                 "final_cache_frames": int(att.shape[2]), "cnn_cache_shape": list(cnn.shape),
                                 "greedy": list(map(int, greedy[0].tokens))})
        arrays["ys_" + key] = ys[0, ::4].numpy().copy()
        arrays["cnn_" + key] = cnn.numpy().copy()
        print("stream", cs, left, tuple(ys.shape), tuple(att.shape), tuple(cnn.shape), len(greedy[0].tokens))
    with open(os.path.join(GOLDEN, c["name"] + ".json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(GOLDEN, c["name"] + ".npz"), **arrays)

    # ---------------------------------------------------------------- plain causal model: the simulate_streaming seam
    c = PLAIN
    x, lens = chunk_feats(fbank_ref.fbank(synth.synth_audio(c["seconds"], seed=1234 + c["seed"])), c["chunk"])
    model, beta = calibrated_model(c, x, lens, None)
    out = {"case": c, "beta": beta, "gamma": synth.CTC_GAMMA, "lens": lens.tolist(), "offline": [], "streaming": []}
    with torch.no_grad():
        res = model.decode(MODES, torch.from_numpy(x), torch.from_numpy(lens), c["beam"], ctc_weight=c["ctc_weight"],
                           reverse_weight=c["reverse_weight"], blank_id=0, infos=infos)
    out["offline"].append({"decoding_chunk_size": -1, "num_decoding_left_chunks": -1, "chunks": rows_of(res)})
    for cs, left in PLAIN_STREAM:
        rows = []
        for b in range(len(lens)):          # the reference's streaming path asserts batch 1
            with torch.no_grad():
                res = model.decode(MODES, torch.from_numpy(x[b:b + 1]), torch.from_numpy(lens[b:b + 1]), c["beam"],
                                   decoding_chunk_size=cs, num_decoding_left_chunks=left, ctc_weight=c["ctc_weight"],
                                   simulate_streaming=True, reverse_weight=c["reverse_weight"], blank_id=0, infos=infos)
            rows += rows_of(res)
        out["streaming"].append({"decoding_chunk_size": cs, "num_decoding_left_chunks": left, "chunks": rows})
        print("plain", cs, left, [len(r["greedy"]) for r in rows], [len(r["rescoring"]) for r in rows])
    with open(os.path.join(GOLDEN, c["name"] + ".json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
