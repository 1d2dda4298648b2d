"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz + *.json by running the UNMODIFIED
reference (/root/reference/asr/wenet, imported through oracle/ref_shim.py) on synthetic weights
and synthetic audio.  Run in the build container:  python -m oracle.gen_golden

The reference cannot travel to the GPU box, so the vectors are committed.  Weights and audio are
NOT stored: they are regenerated bit-identically from (dims, seed, gamma, beta) by
reverb_amd/synth.py, which this script proves compatible by `load_state_dict(strict=True)` into
the reference's own `init_model` graph.
"""
from __future__ import annotations

import copy
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fbank_ref, ref_shim  # noqa: E402
from reverb_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = [
    # name, dims, norm, seed, audio seconds, chunk_size, beam, ctc_weight, reverse_weight, cat_embs, extra short chunk
    dict(name="tiny_ln", dims="tiny", norm="layer_norm", seed=0, seconds=30.0, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.0, cat=[1.0, 0.0]),
    dict(name="tiny_ln_r2l", dims="tiny", norm="layer_norm", seed=1, seconds=12.0, chunk=1000, beam=5, ctc_weight=0.3,
         reverse_weight=0.3, cat=[0.3, 0.7]),
    dict(name="tiny_bn", dims="tiny", norm="batch_norm", seed=2, seconds=9.0, chunk=703, beam=4, ctc_weight=0.1,
         reverse_weight=0.0, cat=[0.0, 1.0], tail_frames=5),
    dict(name="small_ln", dims="small", norm="layer_norm", seed=3, seconds=30.0, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.5, cat=[1.0, 0.0]),
    dict(name="r268_chunk", dims="r268", norm="layer_norm", seed=0, seconds=20.6, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.0, cat=[1.0, 0.0], light=True),
]


# Long-form cases: the reference decodes chunk by chunk with batch 1 exactly as its CLI does (cli/reverb.py:220-253,
# recognize_wav.py:60-64); only the winners are stored (tokens, times, scores) plus a strided encoder sample of the
# first and the last chunk.  `r640_1h` IS the bench workload (bench.py: r640 weights, seed 0, beta frozen in
# synth.CTC_BLANK_BIAS, 1 h of synth_audio(seed=1234), 176 chunks): what the 144 + 32 slice bf16 path is judged on.
LONG_CASES = [
    dict(name="small_66", dims="small", norm="layer_norm", seed=4, seconds=1340.0, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.0, cat=[1.0, 0.0]),
    dict(name="r640_chunk", dims="r640", norm="layer_norm", seed=0, seconds=20.6, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.0, cat=[1.0, 0.0], frozen_beta=True),
    dict(name="r640_1h", dims="r640", norm="layer_norm", seed=0, seconds=3600.0, chunk=2051, beam=10, ctc_weight=0.1,
         reverse_weight=0.0, cat=[1.0, 0.0], frozen_beta=True),
]


class _Args:
    jit = False


def build_reference_model(cfg, sd):
    from wenet.utils.init_model import init_model
    d = tempfile.mkdtemp()
    synth.write_model_dir(d, "x", sd={}, cfg=cfg)
    c2 = copy.deepcopy(cfg)
    c2["cmvn_conf"]["cmvn_file"] = os.path.join(d, "global_cmvn.json")
    model, _ = init_model(_Args(), c2)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    return model, d


def chunk_feats(feats: np.ndarray, chunk: int, tail_frames=None):
    """feats_batcher semantics (cli/reverb.py:148-180) with batch = all chunks."""
    n = feats.shape[0]
    if tail_frames is not None:                 # force a last chunk with very few valid frames
        n = (n // chunk) * chunk + tail_frames
        feats = feats[:n]
    nch = -(-n // chunk)
    x = np.zeros((nch, chunk, feats.shape[1]), np.float32)
    lens = np.full(nch, chunk, np.int32)
    for c in range(nch):
        part = feats[c * chunk:(c + 1) * chunk]
        x[c, :len(part)] = part
        lens[c] = len(part)
    return x, lens


def calibrate_beta(model, x, lens, cat, blank=0):
    """beta = 0.84-quantile of (max non-blank logit - blank logit) on valid frames."""
    with torch.no_grad():
        enc, mask = model.encoder(torch.from_numpy(x[:1]), torch.from_numpy(lens[:1]), -1, -1, cat_embs=cat)
        logits = model.ctc.ctc_lo(enc)[0, :int(mask.sum())]
    nb = logits.clone()
    nb[:, blank] = -1e30
    return float(torch.quantile(nb.max(-1).values - logits[:, blank], 0.84))


def run_case(case):
    from wenet.cli.reverb import get_output
    from wenet.text.rev_bpe_tokenizer import RevBpeTokenizer
    cfg = synth.make_config(case["dims"], case["norm"])
    pcm = synth.synth_audio(case["seconds"], seed=1234 + case["seed"])
    feats = fbank_ref.fbank(pcm)
    x, lens = chunk_feats(feats, case["chunk"], case.get("tail_frames"))
    cat = torch.tensor(case["cat"])
    # pass 1: un-biased head to measure beta, pass 2: calibrated weights
    sd0 = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, 0.0)
    model, _ = build_reference_model(cfg, sd0)
    beta = calibrate_beta(model, x, lens, cat)
    sd = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, beta)
    model, mdir = build_reference_model(cfg, sd)
    modes = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]
    with torch.no_grad():
        res = model.decode(modes, torch.from_numpy(x), torch.from_numpy(lens), case["beam"], ctc_weight=case["ctc_weight"],
                           reverse_weight=case["reverse_weight"], cat_embs=cat, blank_id=0,
                           infos={"tasks": ["transcribe"], "langs": ["en"]})
        enc, mask = model.encoder(torch.from_numpy(x), torch.from_numpy(lens), -1, -1, cat_embs=cat)
        probs = model.ctc_logprobs(enc)
        tv, ti = probs.topk(case["beam"], dim=-1)
    units = synth.make_units(cfg["output_dim"])
    tok = RevBpeTokenizer(None, {u: i for i, u in enumerate(units)})
    outputs = {}
    for m in ("ctc_prefix_beam_search", "attention_rescoring"):
        try:
            outputs[m] = {fmt: get_output(fmt, tok, "golden.wav", res[m], 230, case["chunk"], 10, 40) for fmt in ("ctm", "txt")}
        except AssertionError as ex:   # e.g. times shorter than tokens
            outputs[m] = {"error": "AssertionError"}
    js = dict(case=case, beta=beta, gamma=synth.CTC_GAMMA, encoder_lens=mask.squeeze(1).sum(1).tolist(),
              lens=lens.tolist(), outputs=outputs, modes={})
    for m in modes:
        js["modes"][m] = [dict(tokens=list(map(int, r.tokens)), score=float(r.score), confidence=float(r.confidence),
                               tokens_confidence=None if r.tokens_confidence is None else [float(c) for c in r.tokens_confidence],
                               times=None if r.times is None else list(map(int, r.times)),
                               nbest=None if r.nbest is None else [list(map(int, h)) for h in r.nbest],
                               nbest_scores=None if r.nbest_scores is None else [float(s) for s in r.nbest_scores],
                               nbest_times=None if r.nbest_times is None else [list(map(int, t)) for t in r.nbest_times])
                          for r in res[m]]
    arrays = dict(topk_val=tv.numpy(), topk_idx=ti.numpy().astype(np.int32))
    if case.get("light"):
        arrays["encoder_out"] = enc.numpy()[:, ::16, ::8].copy()      # strided sample keeps the fixture small
    else:
        arrays["encoder_out"] = enc.numpy()
        arrays["ctc_probs_chunk0"] = probs[0].numpy() if cfg["output_dim"] <= 64 else probs[0, ::8].numpy()
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, case["name"] + ".npz"), **arrays)
    with open(os.path.join(GOLDEN, case["name"] + ".json"), "w") as f:
        json.dump(js, f, indent=1)
    ntok = [len(r.tokens) for r in res["attention_rescoring"]]
    print(f"{case['name']}: beta={beta:.4f} chunks={len(lens)} enc_lens={js['encoder_lens']} rescored tokens/chunk={ntok}")


def run_long_case(case):
    import time
    cfg = synth.make_config(case["dims"], case["norm"])
    pcm = synth.synth_audio(case["seconds"], seed=1234 + case["seed"])
    feats = fbank_ref.fbank(pcm)
    x, lens = chunk_feats(feats, case["chunk"])
    cat = torch.tensor(case["cat"])
    if case.get("frozen_beta"):      # the weights bench.py runs: synth.calibrated_state_dict(dims, seed)
        beta = synth.CTC_BLANK_BIAS[(case["dims"], case["seed"])]
    else:
        model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, 0.0))
        beta = calibrate_beta(model, x, lens, cat)
        del model
    sd = synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, beta)
    model, _ = build_reference_model(cfg, sd)
    del sd
    modes = ["ctc_greedy_search", "attention_rescoring"]
    rows = {m: [] for m in modes}
    enc_lens, arrays = [], {}
    t0 = time.time()
    for c in range(len(lens)):
        xc, lc = torch.from_numpy(x[c:c + 1]), torch.from_numpy(lens[c:c + 1])
        with torch.no_grad():
            res = model.decode(modes, xc, lc, case["beam"], ctc_weight=case["ctc_weight"], reverse_weight=case["reverse_weight"],
                               cat_embs=cat, blank_id=0, infos={"tasks": ["transcribe"], "langs": ["en"]})
            if c in (0, len(lens) - 1):
                enc, mask = model.encoder(xc, lc, -1, -1, cat_embs=cat)
                probs = model.ctc_logprobs(enc)
                n = int(mask.sum())
                arrays[f"encoder_out_{c}"] = enc[0, :n:16, ::8].numpy().copy()
                arrays[f"ctc_top1_{c}"] = probs[0, :n].max(-1).values.numpy().copy()
                arrays[f"ctc_argmax_{c}"] = probs[0, :n].argmax(-1).numpy().astype(np.int32)
        g, r = res["ctc_greedy_search"][0], res["attention_rescoring"][0]
        rows["ctc_greedy_search"].append(dict(tokens=list(map(int, g.tokens))))
        rows["attention_rescoring"].append(dict(tokens=list(map(int, r.tokens)), times=list(map(int, r.times)),
                                                score=float(r.score), confidence=float(r.confidence)))
        enc_lens.append(int(lc[0] > 6) * ((int(lc[0]) - 7) // 4 + 1))
        if c % 8 == 0:
            print(f"  {case['name']}: chunk {c + 1}/{len(lens)}  {time.time() - t0:.0f} s", flush=True)
    js = dict(case=dict(case, long=True), beta=beta, gamma=synth.CTC_GAMMA, lens=lens.tolist(), encoder_lens=enc_lens, modes=rows)
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, case["name"] + ".npz"), **arrays)
    with open(os.path.join(GOLDEN, case["name"] + ".json"), "w") as f:
        json.dump(js, f, separators=(",", ":"))
    ntok = sum(len(r["tokens"]) for r in rows["attention_rescoring"])
    print(f"{case['name']}: beta={beta:.4f} chunks={len(lens)} rescored tokens={ntok}  {time.time() - t0:.0f} s")


def fbank_golden():
    """Pin the fbank restatement with the independent Kaldi-compatible implementation shipped in
    `transformers` (the reference's own torchaudio is not installable here: parity unpinned)."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    pcm = synth.synth_audio(2.0, seed=99)
    mf = mel_filter_bank(257, 80, 20, 8000, 16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    g = spectrogram(pcm.astype(np.float32), window_function(400, "povey", periodic=False), frame_length=400,
                    hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=mf,
                    log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    np.savez_compressed(os.path.join(GOLDEN, "fbank_transformers.npz"), feats=g.astype(np.float32))
    print("fbank golden", g.shape, "max |ours-transformers| =", float(np.abs(fbank_ref.fbank(pcm) - g).max()))


if __name__ == "__main__":
    only = sys.argv[1:]
    torch.set_num_threads(8)
    if not only or "fbank" in only:
        fbank_golden()          # before the shim: its torchaudio stand-in confuses transformers' import probe
    ref_shim.install()
    for case in CASES:
        if only and case["name"] not in only:
            continue
        run_case(case)
    for case in LONG_CASES:          # long cases only on request: r640_1h takes ~20 min of CPU
        if case["name"] in only:
            run_long_case(case)
