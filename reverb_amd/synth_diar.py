"""Synthetic weights of the diarization networks, under the state-dict names of the pyannote
checkpoints the reference downloads (`Revai/reverb-diarization-v1`: a fine-tuned
pyannote/segmentation-3.0 plus pyannote/wespeaker-voxceleb-resnet34-LM; there is no network here,
so neither is available).  Shapes are those of the published architectures; values are seeded
random draws scaled so that activations stay O(1) through every layer and the powerset classes
are all visited (bench.py and the parity tests use the same generator).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

DIAR_DIMS = dict(sample_rate=16000, window_samples=160000, step_samples=16000, sinc_filters=80, sinc_channels=60,
                 lstm_hidden=128, lstm_layers=4, linear_dim=128, linear_layers=2, num_classes=7,
                 emb_channels=32, emb_dim=256)


def make_diar_config(**over) -> dict:
    cfg = dict(DIAR_DIMS)
    cfg.update(over)
    return cfg


def _u(rng, shape, k):
    return rng.uniform(-k, k, size=shape).astype(np.float32)


def make_segmentation_sd(cfg: dict = None, seed: int = 0) -> Dict[str, np.ndarray]:
    """PyanNet (SincNet + BiLSTM + linears + powerset classifier) state dict."""
    cfg = cfg or DIAR_DIMS
    rng = np.random.default_rng(seed + 7000)
    sd = {}
    sd["sincnet.wav_norm1d.weight"] = np.array([1.0 + 0.1 * rng.standard_normal()], np.float32)
    sd["sincnet.wav_norm1d.bias"] = np.array([0.05 * rng.standard_normal()], np.float32)
    # asteroid ParamSincFB initialisation: band edges equally spaced on the mel scale
    npair = cfg["sinc_filters"] // 2
    to_mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
    to_hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    hz = to_hz(np.linspace(to_mel(30.0), to_mel(cfg["sample_rate"] / 2 - 100.0), npair + 1))
    sd["sincnet.conv1d.0.filterbank.low_hz_"] = hz[:-1].astype(np.float32).reshape(-1, 1)
    sd["sincnet.conv1d.0.filterbank.band_hz_"] = np.diff(hz).astype(np.float32).reshape(-1, 1)
    C = cfg["sinc_channels"]
    for i, cin in ((1, cfg["sinc_filters"]), (2, C)):
        k = 1.0 / math.sqrt(cin * 5)
        sd[f"sincnet.conv1d.{i}.weight"] = _u(rng, (C, cin, 5), 1.7 * k)
        sd[f"sincnet.conv1d.{i}.bias"] = _u(rng, (C,), k)
    for i, ch in ((0, cfg["sinc_filters"]), (1, C), (2, C)):
        sd[f"sincnet.norm1d.{i}.weight"] = (1.0 + 0.1 * rng.standard_normal(ch)).astype(np.float32)
        sd[f"sincnet.norm1d.{i}.bias"] = (0.1 * rng.standard_normal(ch)).astype(np.float32)
    H = cfg["lstm_hidden"]
    for layer in range(cfg["lstm_layers"]):
        cin = C if layer == 0 else 2 * H
        for suf in ("", "_reverse"):
            sd[f"lstm.weight_ih_l{layer}{suf}"] = _u(rng, (4 * H, cin), 4.0 / math.sqrt(cin))
            sd[f"lstm.weight_hh_l{layer}{suf}"] = _u(rng, (4 * H, H), 3.0 / math.sqrt(H))
            sd[f"lstm.bias_ih_l{layer}{suf}"] = _u(rng, (4 * H,), 1.0 / math.sqrt(H))
            sd[f"lstm.bias_hh_l{layer}{suf}"] = _u(rng, (4 * H,), 1.0 / math.sqrt(H))
    cin = 2 * H
    for i in range(cfg["linear_layers"]):
        sd[f"linear.{i}.weight"] = _u(rng, (cfg["linear_dim"], cin), 4.0 / math.sqrt(cin))
        sd[f"linear.{i}.bias"] = _u(rng, (cfg["linear_dim"],), 1.0 / math.sqrt(cin))
        cin = cfg["linear_dim"]
    sd["classifier.weight"] = _u(rng, (cfg["num_classes"], cin), 12.0 / math.sqrt(cin))
    sd["classifier.bias"] = _u(rng, (cfg["num_classes"],), 0.5)
    return sd


def make_embedding_sd(cfg: dict = None, seed: int = 0) -> Dict[str, np.ndarray]:
    """WeSpeaker ResNet34 (BasicBlock [3,4,6,3], m_channels 32, feat_dim 80, TSTP, embed_dim 256)."""
    cfg = cfg or DIAR_DIMS
    rng = np.random.default_rng(seed + 9000)
    m = cfg["emb_channels"]
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * math.sqrt(2.0 / (cin * k * k))).astype(np.float32)

    def bn(name, ch, lo=0.6, hi=1.2):
        sd[name + ".weight"] = rng.uniform(lo, hi, ch).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rng.standard_normal(ch)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rng.standard_normal(ch)).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, ch).astype(np.float32)

    conv("resnet.conv1", m, 1, 3); bn("resnet.bn1", m)
    cin = m
    for li, (nblk, stride) in enumerate(zip((3, 4, 6, 3), (1, 2, 2, 2)), start=1):
        cout = m * (2 ** (li - 1))
        for b in range(nblk):
            p = f"resnet.layer{li}.{b}"
            s = stride if b == 0 else 1
            conv(p + ".conv1", cout, cin, 3); bn(p + ".bn1", cout)
            conv(p + ".conv2", cout, cout, 3); bn(p + ".bn2", cout, 0.15, 0.4)   # small residual branches, as in trained nets
            if s != 1 or cin != cout:
                conv(p + ".shortcut.0", cout, cin, 1); bn(p + ".shortcut.1", cout)
            cin = cout
    stats = (80 // 8) * m * 8 * 2
    sd["resnet.seg_1.weight"] = _u(rng, (cfg["emb_dim"], stats), 1.0 / math.sqrt(stats))
    sd["resnet.seg_1.bias"] = _u(rng, (cfg["emb_dim"],), 1.0 / math.sqrt(stats))
    return sd


def synth_conversation(seconds: float, seed: int = 4321, sample_rate: int = 16000, speakers: int = 3) -> np.ndarray:
    """int16 mono 'conversation': turns of 1-6 s by `speakers` voices that differ in pitch and formant
    placement, with pauses and some overlapped turns -- enough structure for the windowed statistics
    (instance norms, masks, clustering) to be exercised; not speech."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sample_rate)
    out = np.zeros(n, np.float64)
    t = 0
    f0s = [95.0, 140.0, 205.0, 170.0, 120.0][:speakers]
    while t < n:
        spk = int(rng.integers(speakers))
        dur = int(rng.uniform(1.0, 6.0) * sample_rate)
        seg = np.arange(min(dur, n - t)) / sample_rate
        f0 = f0s[spk] * (1.0 + 0.03 * np.sin(2 * np.pi * 3.1 * seg + spk))
        ph = 2 * np.pi * np.cumsum(f0) / sample_rate
        v = np.zeros_like(seg)
        for h in range(1, 24):
            fh = h * f0s[spk]
            form = sum(np.exp(-0.5 * ((fh - fc) / bw) ** 2) for fc, bw in ((500 + 120 * spk, 130), (1500 + 200 * spk, 220), (2600 + 90 * spk, 300)))
            v += (0.05 + form) / h ** 0.5 * np.sin(h * ph + 0.3 * h)
        env = 0.55 + 0.45 * np.sin(2 * np.pi * rng.uniform(2.5, 4.5) * seg + rng.uniform(0, 6.28)) ** 2
        v = v * env * rng.uniform(0.5, 1.0) + 0.02 * rng.standard_normal(seg.shape[0])
        out[t:t + seg.shape[0]] += v
        gap = rng.uniform(-1.0, 1.2)          # negative: the next turn overlaps this one
        t += max(int((dur / sample_rate + gap) * sample_rate), sample_rate // 4)
    out += 0.003 * rng.standard_normal(n)
    out = out / (np.abs(out).max() + 1e-9) * 0.7
    return np.round(out * 32767.0).astype(np.int16)


def write_pipeline_dir(path: str, seed: int = 0, cfg: dict = None) -> str:
    """A local pipeline directory in the layout `reverb_amd.diarization.Pipeline.from_pretrained` reads:
    config.yaml (pyannote/speaker-diarization-3.1 hyper-parameters) + segmentation.pt + embedding.pt."""
    import os
    import torch
    import yaml
    cfg = cfg or DIAR_DIMS
    os.makedirs(path, exist_ok=True)
    conf = {
        "version": "3.1.0",
        "pipeline": {"name": "pyannote.audio.pipelines.SpeakerDiarization",
                     "params": {"clustering": "AgglomerativeClustering", "embedding": "embedding.pt", "embedding_batch_size": 32,
                                "embedding_exclude_overlap": True, "segmentation": "segmentation.pt", "segmentation_batch_size": 32}},
        "params": {"clustering": {"method": "centroid", "min_cluster_size": 12, "threshold": 0.7045654963945799},
                   "segmentation": {"min_duration_off": 0.0}},
    }
    with open(os.path.join(path, "config.yaml"), "w") as f:
        yaml.safe_dump(conf, f)
    torch.save({k: torch.from_numpy(v) for k, v in make_segmentation_sd(cfg, seed).items()}, os.path.join(path, "segmentation.pt"))
    torch.save({k: torch.from_numpy(v) for k, v in make_embedding_sd(cfg, seed).items()}, os.path.join(path, "embedding.pt"))
    return path
