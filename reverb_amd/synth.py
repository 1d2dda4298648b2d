"""Synthetic Reverb-ASR model directories, tokenizer tables and audio.

The real artefacts (`config.yaml`, `reverb_asr_v1.pt`, `tk.units.txt`, the CMVN
file) are fetched by the reference from HuggingFace
(`asr/wenet/cli/reverb.py:40-43, 330-378`); there is no network here, so
benchmarks, smoke tests and parity tests run on *synthetic* weights that have
exactly the reference's state-dict names and shapes (SURVEY.md §8 row a-W,
verified by loading them into the reference's own `init_model` graph with
`strict=True` in `oracle/gen_golden.py`).

Everything is generated from numpy `default_rng(seed)` so the same bytes are
reproduced in the build container (where the reference is imported to create
golden vectors) and on the GPU box (where `/root/reference` does not exist).
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict

import numpy as np

# ---------------------------------------------------------------------------
# planning points (SURVEY.md §8 header)
# ---------------------------------------------------------------------------
MODEL_DIMS = {
    # name: d, heads, ff, enc blocks, dec-ff, dec blocks (each direction), cnn kernel, vocab
    "tiny":  dict(d=32,   h=2,  ff=64,   blocks=4,  dff=64,   dblocks=3, K=15, vocab=48),
    "small": dict(d=128,  h=4,  ff=256,  blocks=4,  dff=256,  dblocks=3, K=15, vocab=500),
    # the Reverb vocabulary size on small bodies: `joint_decoding` hard-codes sos = 10000 (search.py:479)
    "tiny_v10k":  dict(d=32,  h=2, ff=64,  blocks=4, dff=64,  dblocks=3, K=15, vocab=10001),
    "small_v10k": dict(d=128, h=4, ff=256, blocks=4, dff=256, dblocks=3, K=15, vocab=10001),
    "r268":  dict(d=640,  h=8,  ff=2560, blocks=18, dff=2048, dblocks=3, K=31, vocab=10001),
    "r640":  dict(d=1024, h=16, ff=4096, blocks=18, dff=4096, dblocks=3, K=31, vocab=10001),
}

# CMVN statistics of the synthetic audio below (mean / istd of its 80-bin log-mel),
# measured once with oracle/fbank_ref.py on 60 s of `synth_audio(seed=1234)` and frozen
# (oracle/calibrate.py regenerates them).  Frozen so that product code never needs the oracle.
_CMVN_MEAN = None  # filled in below
_CMVN_ISTD = None

# CTC-head calibration (SURVEY.md §8d "Synthetic model"): gamma scales ctc_lo, beta is added to
# the blank bias so that greedy emits a speech-like ~4 tokens/s.  Measured by
# oracle/calibrate.py per (dims, seed) and frozen here.
CTC_GAMMA = 8.0
CTC_BLANK_BIAS = {
    # (name, seed): beta   -- python -m oracle.calibrate tiny small r268 r640
    ("tiny", 0): 12.3299,
    ("small", 0): 17.8826,
    ("r268", 0): 14.2324,
    ("r640", 0): 24.3962,
}


def make_config(name: str = "r640", cnn_module_norm: str = "layer_norm", causal: bool = False,
                use_dynamic_chunk: bool = False, cnn_module_kernel: int = None, pass_cat_emb: bool = True) -> dict:
    """Reverb-ASR style config.yaml contents (keys consumed by
    `asr/wenet/utils/init_model.py:102-183,252-265` and `cli/reverb.py:62-98`).  `causal` + `use_dynamic_chunk` give
    the stock WeNet U2++ streaming recipe (causal convolution module, dynamic-chunk attention)."""
    m = dict(MODEL_DIMS[name])
    if cnn_module_kernel is not None:
        m["K"] = int(cnn_module_kernel)
    cfg = _make_config(m, cnn_module_norm)
    cfg["encoder_conf"]["causal"] = bool(causal)
    if use_dynamic_chunk:
        cfg["encoder_conf"]["use_dynamic_chunk"] = True
    if not pass_cat_emb:
        cfg["dataset_conf"]["pass_cat_emb"] = False
    return cfg


def _make_config(m: dict, cnn_module_norm: str) -> dict:
    return {
        "model": "asr_model",
        "encoder": "conformer",
        "decoder": "bitransformer",
        "ctc": "ctc",
        "cmvn": "global_cmvn",
        "cmvn_conf": {"cmvn_file": "global_cmvn.json", "is_json_cmvn": True},
        "input_dim": 80,
        "output_dim": m["vocab"],
        "tokenizer": "rev_bpe",
        "tokenizer_conf": {
            "symbol_table_path": "tk.units.txt",
            "bpe_path": "tk.model",
            "non_lang_syms_path": None,
            "split_with_space": False,
            "special_tokens": None,
        },
        "ctc_conf": {"ctc_blank_id": 0},
        "encoder_conf": {
            "output_size": m["d"],
            "attention_heads": m["h"],
            "linear_units": m["ff"],
            "num_blocks": m["blocks"],
            "dropout_rate": 0.1,
            "positional_dropout_rate": 0.1,
            "attention_dropout_rate": 0.1,
            "input_layer": "conv2d",
            "pos_enc_layer_type": "rel_pos",
            "selfattention_layer_type": "rel_selfattn",
            "activation_type": "swish",
            "normalize_before": True,
            "use_cnn_module": True,
            "cnn_module_kernel": m["K"],
            "cnn_module_norm": cnn_module_norm,
            "causal": False,
            "macaron_style": True,
        },
        "decoder_conf": {
            "attention_heads": m["h"],
            "linear_units": m["dff"],
            "num_blocks": m["dblocks"],
            "r_num_blocks": m["dblocks"],
            "dropout_rate": 0.1,
            "positional_dropout_rate": 0.1,
            "self_attention_dropout_rate": 0.1,
            "src_attention_dropout_rate": 0.1,
        },
        "model_conf": {
            "ctc_weight": 0.3,
            "lsm_weight": 0.1,
            "reverse_weight": 0.3,
            "length_normalized_loss": False,
        },
        "dataset_conf": {
            "pass_cat_emb": True,
            "cat_emb_conf": {"emb_len": 2, "one_hot_ids": {"vb": 0, "nv": 1}},
            "fbank_conf": {"num_mel_bins": 80, "frame_length": 25, "frame_shift": 10, "dither": 0.0},
        },
    }


# ---------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------
def _lin(rng, out_f, in_f, bias=True, prefix="", sd=None):
    b = 1.0 / math.sqrt(in_f)
    sd[prefix + ".weight"] = rng.uniform(-b, b, size=(out_f, in_f)).astype(np.float32)
    if bias:
        sd[prefix + ".bias"] = rng.uniform(-b, b, size=(out_f,)).astype(np.float32)


def _ln(rng, d, prefix, sd):
    sd[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    sd[prefix + ".bias"] = (0.05 * rng.standard_normal(d)).astype(np.float32)


def _mha(rng, d, prefix, sd):
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        _lin(rng, d, d, True, f"{prefix}.{n}", sd)


def make_state_dict(cfg: dict, seed: int = 0, ctc_gamma: float = 1.0,
                    ctc_blank_bias: float = 0.0) -> Dict[str, np.ndarray]:
    """All tensors of the reference's `ASRModel.state_dict()` for `cfg`
    (names: SURVEY.md §8 row a-W), as float32 numpy arrays, in the reference's
    registration order."""
    rng = np.random.default_rng(seed)
    ec, dc = cfg["encoder_conf"], cfg["decoder_conf"]
    d, h, ff = ec["output_size"], ec["attention_heads"], ec["linear_units"]
    K = ec["cnn_module_kernel"]
    V = cfg["output_dim"]
    idim = cfg["input_dim"]
    nlang = cfg["dataset_conf"]["cat_emb_conf"]["emb_len"] if cfg["dataset_conf"].get("pass_cat_emb") else 0
    dk = d // h
    sd: Dict[str, np.ndarray] = {}

    mean, istd = cmvn_stats()
    sd["encoder.global_cmvn.mean"] = mean.astype(np.float32)
    sd["encoder.global_cmvn.istd"] = istd.astype(np.float32)
    # Conv2dSubsampling4 (subsampling.py:172-199)
    b = 1.0 / math.sqrt(9.0)
    sd["encoder.embed.conv.0.weight"] = rng.uniform(-b, b, size=(d, 1, 3, 3)).astype(np.float32)
    sd["encoder.embed.conv.0.bias"] = rng.uniform(-b, b, size=(d,)).astype(np.float32)
    b = 1.0 / math.sqrt(9.0 * d)
    sd["encoder.embed.conv.2.weight"] = rng.uniform(-b, b, size=(d, d, 3, 3)).astype(np.float32)
    sd["encoder.embed.conv.2.bias"] = rng.uniform(-b, b, size=(d,)).astype(np.float32)
    fdim = ((idim - 1) // 2 - 1) // 2
    _lin(rng, d, d * fdim, True, "encoder.embed.out.0", sd)
    _ln(rng, d, "encoder.after_norm", sd)
    nb = ec["num_blocks"]
    for i in range(nb):
        p = f"encoder.encoders.{i}"
        lsl = nlang > 0 and i in (0, nb - 1)
        _mha(rng, d, p + ".self_attn", sd)
        _lin(rng, d, d, False, p + ".self_attn.linear_pos", sd)
        xb = math.sqrt(6.0 / (h + dk))
        # registration order in the reference: pos_bias_u/v are Parameters of the attention
        sd[p + ".self_attn.pos_bias_u"] = rng.uniform(-xb, xb, size=(h, dk)).astype(np.float32)
        sd[p + ".self_attn.pos_bias_v"] = rng.uniform(-xb, xb, size=(h, dk)).astype(np.float32)
        for f in ("feed_forward", "feed_forward_macaron"):
            _lin(rng, ff, d, True, f"{p}.{f}.w_1", sd)
            _lin(rng, d, ff, True, f"{p}.{f}.w_2", sd)
        cb = 1.0 / math.sqrt(d)
        sd[p + ".conv_module.pointwise_conv1.weight"] = rng.uniform(-cb, cb, size=(2 * d, d, 1)).astype(np.float32)
        sd[p + ".conv_module.pointwise_conv1.bias"] = rng.uniform(-cb, cb, size=(2 * d,)).astype(np.float32)
        kb = 1.0 / math.sqrt(K)
        sd[p + ".conv_module.depthwise_conv.weight"] = rng.uniform(-kb, kb, size=(d, 1, K)).astype(np.float32)
        sd[p + ".conv_module.depthwise_conv.bias"] = rng.uniform(-kb, kb, size=(d,)).astype(np.float32)
        _ln(rng, d, p + ".conv_module.norm", sd)
        if ec.get("cnn_module_norm", "batch_norm") == "batch_norm":
            sd[p + ".conv_module.norm.running_mean"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
            sd[p + ".conv_module.norm.running_var"] = rng.uniform(0.5, 1.5, size=(d,)).astype(np.float32)
            sd[p + ".conv_module.norm.num_batches_tracked"] = np.array(1, dtype=np.int64)
        sd[p + ".conv_module.pointwise_conv2.weight"] = rng.uniform(-cb, cb, size=(d, d, 1)).astype(np.float32)
        sd[p + ".conv_module.pointwise_conv2.bias"] = rng.uniform(-cb, cb, size=(d,)).astype(np.float32)
        for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            _ln(rng, d, f"{p}.{n}", sd)
        if lsl:
            for j in range(nlang):
                _lin(rng, d, d, True, f"{p}.language_layers.{j}", sd)

    dff = dc["linear_units"]
    for side, nblk in (("left_decoder", dc["num_blocks"]), ("right_decoder", dc["r_num_blocks"])):
        p = f"decoder.{side}"
        sd[p + ".embed.0.weight"] = rng.standard_normal((V, d)).astype(np.float32)
        _ln(rng, d, p + ".after_norm", sd)
        _lin(rng, V, d, True, p + ".output_layer", sd)
        for j in range(nblk):
            q = f"{p}.decoders.{j}"
            lsl = nlang > 0 and j in (0, nblk - 1)
            _mha(rng, d, q + ".self_attn", sd)
            _mha(rng, d, q + ".src_attn", sd)
            _lin(rng, dff, d, True, q + ".feed_forward.w_1", sd)
            _lin(rng, d, dff, True, q + ".feed_forward.w_2", sd)
            for n in ("norm1", "norm2", "norm3"):
                _ln(rng, d, f"{q}.{n}", sd)
            if lsl:
                # dead weights present in the checkpoint (decoder_layer.py:246-247)
                _lin(rng, d, 2 * d, True, q + ".concat_linear1", sd)
                _lin(rng, d, 2 * d, True, q + ".concat_linear2", sd)
                for jj in range(nlang):
                    _lin(rng, d, d, True, f"{q}.language_layers.{jj}", sd)
    _lin(rng, V, d, True, "ctc.ctc_lo", sd)
    sd["ctc.ctc_lo.weight"] *= np.float32(ctc_gamma)
    sd["ctc.ctc_lo.bias"] *= np.float32(ctc_gamma)
    sd["ctc.ctc_lo.bias"][cfg["ctc_conf"]["ctc_blank_id"]] += np.float32(ctc_blank_bias)
    return sd


def calibrated_state_dict(name: str, seed: int = 0, cnn_module_norm: str = "layer_norm"):
    cfg = make_config(name, cnn_module_norm)
    beta = CTC_BLANK_BIAS.get((name, seed))
    if beta is None:
        raise KeyError(f"no frozen CTC calibration for ({name!r}, {seed}); run oracle/calibrate.py")
    return cfg, make_state_dict(cfg, seed, CTC_GAMMA, beta)


# ---------------------------------------------------------------------------
# tokenizer table, cmvn, audio, model dir
# ---------------------------------------------------------------------------
_SYL = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "pe", "ho", "da", "qu", "ze", "ba", "ny", "fi"]


def make_units(vocab: int):
    """id -> piece list: `<blank> 0`, `<unk> 1`, pieces (every other one word-initial `▁..`),
    a few `<tag>` special tokens, `<sos/eos> vocab-1` (asr_model.py:79-82)."""
    units = ["<blank>", "<unk>"]
    i = 0
    while len(units) < vocab - 1:
        k = len(units)
        if k % 97 == 5:
            units.append(f"<tag{k}>")
        else:
            a, b, c = _SYL[i % 16], _SYL[(i // 16) % 16], i // 256
            piece = a + b + (str(c) if c else "")
            units.append(("▁" + piece) if (i % 2 == 0) else piece)
            i += 1
    units.append("<sos/eos>")
    return units


def synth_audio(seconds: float, seed: int = 1234, sample_rate: int = 16000) -> np.ndarray:
    """16 kHz mono int16: FM tone + noise with a 4 Hz raised-cosine envelope."""
    n = int(round(seconds * sample_rate))
    rng = np.random.default_rng(seed)
    out = np.empty(n, dtype=np.int16)
    blk = 1 << 22
    for s in range(0, n, blk):
        e = min(n, s + blk)
        t = np.arange(s, e, dtype=np.float64) / sample_rate
        x = 0.6 * np.sin(2 * np.pi * (220.0 + 110.0 * np.sin(2 * np.pi * 0.25 * t)) * t)
        x += 0.1 * rng.standard_normal(e - s)
        env = 0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * t)
        x = x * (0.15 + 0.85 * env)
        out[s:e] = np.clip(x * 0.5 * 32767.0, -32768, 32767).astype(np.int16)
    return out


def write_wav(path: str, pcm: np.ndarray, sample_rate: int = 16000, channels: int = 1):
    """Minimal RIFF/WAVE writer (44-byte header, PCM16)."""
    import struct
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, sample_rate,
                                      sample_rate * channels * 2, channels * 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


def cmvn_stats():
    return np.asarray(_CMVN_MEAN, np.float64), np.asarray(_CMVN_ISTD, np.float64)


def write_model_dir(path: str, name: str = "r640", seed: int = 0, sd=None, cfg=None):
    """Write `config.yaml`, `<name>.pt`, `tk.units.txt`, `global_cmvn.json` the way
    `load_model(dir)` expects them (cli/reverb.py:339-342)."""
    import torch
    import yaml
    os.makedirs(path, exist_ok=True)
    if cfg is None or sd is None:
        cfg, sd = calibrated_state_dict(name, seed)
    with open(os.path.join(path, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    units = make_units(cfg["output_dim"])
    with open(os.path.join(path, "tk.units.txt"), "w", encoding="utf8") as f:
        for i, u in enumerate(units):
            f.write(f"{u} {i}\n")
    mean, istd = cmvn_stats()
    # json cmvn holds sufficient statistics (utils/cmvn.py:21-43): pick frame_num=1e6
    cnt = 1000000.0
    var = 1.0 / (istd ** 2)
    with open(os.path.join(path, "global_cmvn.json"), "w") as f:
        json.dump({"mean_stat": (mean * cnt).tolist(),
                   "var_stat": ((var + mean * mean) * cnt).tolist(),
                   "frame_num": cnt}, f)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()},
               os.path.join(path, f"{name}.pt"))
    return path


_CMVN_MEAN = [11.102, 11.569, 11.951, 12.33, 12.696, 12.982, 13.2, 13.355, 13.633, 13.826, 14.103, 14.339, 14.59, 14.816, 14.947, 15.13, 15.289, 15.429, 15.755, 15.808, 15.917, 16.287, 16.263, 16.546, 16.625, 16.829, 16.838, 17.003, 17.087, 17.213, 17.379, 17.525, 17.663, 17.803, 17.959, 18.101, 18.228, 18.291, 18.433, 18.462, 18.645, 18.695, 18.873, 18.999, 19.13, 19.239, 19.316, 19.382, 19.496, 19.621, 19.733, 19.871, 19.94, 20.016, 20.09, 20.22, 20.346, 20.455, 20.514, 20.545, 20.682, 20.815, 20.899, 20.954, 21.015, 21.145, 21.22, 21.279, 21.328, 21.438, 21.497, 21.553, 21.679, 21.779, 21.782, 21.865, 21.984, 21.98, 22.041, 22.111]
_CMVN_ISTD = [0.4945, 0.4566, 0.4498, 0.4653, 0.4665, 0.4665, 0.4597, 0.4448, 0.4487, 0.4558, 0.4515, 0.4488, 0.4605, 0.4802, 0.4946, 0.4933, 0.4914, 0.4889, 0.4975, 0.4932, 0.4842, 0.4752, 0.4628, 0.46, 0.4644, 0.4704, 0.4846, 0.5007, 0.5103, 0.5084, 0.509, 0.5059, 0.5039, 0.5032, 0.4977, 0.4902, 0.4898, 0.5076, 0.5301, 0.5347, 0.5367, 0.5364, 0.531, 0.5253, 0.5163, 0.5251, 0.5461, 0.5558, 0.564, 0.5647, 0.5475, 0.5345, 0.5464, 0.5743, 0.5877, 0.5773, 0.5507, 0.5428, 0.5674, 0.5894, 0.5815, 0.5553, 0.5461, 0.5767, 0.5849, 0.5663, 0.5528, 0.5875, 0.5952, 0.5603, 0.563, 0.6106, 0.5752, 0.5457, 0.59, 0.5785, 0.5487, 0.581, 0.5529, 0.54]
