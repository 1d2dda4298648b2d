"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

PARITY UNPINNED.  The arithmetic of the diarization path lives in third-party packages that are
neither vendored under /root/reference nor installed here: `pyannote.audio==3.3.1`
(/root/reference/diarization/requirements.txt:1; call sites infer_pyannote3.0.py:33-42) and, through it,
`asteroid-filterbanks` (ParamSincFB) and WeSpeaker's ResNet34.  The reference has no test that pins
any diarization output.  This file restates the published architectures from memory of those
packages' sources; it is what the HIP kernels are checked against, on
synthetic weights with the packages' own state-dict names:

  segmentation  pyannote.audio.models.segmentation.PyanNet (= pyannote/segmentation-3.0, which
                reverb-diarization-v1 fine-tunes: diarization/train_pyannote3.0.py:42-44)
                  sincnet.wav_norm1d, sincnet.conv1d.0.filterbank.{low_hz_,band_hz_},
                  sincnet.conv1d.{1,2}, sincnet.norm1d.{0,1,2}, lstm.*, linear.{0,1}, classifier
  embedding     pyannote.audio.models.embedding.WeSpeakerResNet34 (wespeaker ResNet34, TSTP pooling)
                  resnet.conv1, resnet.bn1, resnet.layer{1..4}.{i}.{conv1,bn1,conv2,bn2,shortcut.{0,1}},
                  resnet.seg_1
  pipeline      pyannote.audio.pipelines.SpeakerDiarization (3.1 defaults): 10 s windows every 1 s,
                powerset -> multilabel, speaker count, masked embeddings, centroid-linkage
                agglomerative clustering, reconstruction, binarisation -> RTTM.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
SAMPLE_RATE = 16000
WINDOW = 160000          # 10 s
STEP = 16000             # 1 s  (segmentation_step = 0.1 of the window)
NUM_FRAMES = 589         # SincNet frames of a 10 s window
FRAME_STEP = 270         # samples between frames (10 * 3 * 3 * 3)
FRAME_SIZE = 991         # receptive field in samples

POWERSET = [(), (0,), (1,), (2,), (0, 1), (0, 2), (1, 2)]   # 3 speakers, at most 2 at once


def to_torch_sd(sd) -> SD:
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


# ------------------------------------------------------------------------------------ SincNet
def sinc_filters(low_hz_: torch.Tensor, band_hz_: torch.Tensor, kernel_size=251, sample_rate=16000.0,
                 min_low_hz=50.0, min_band_hz=50.0) -> torch.Tensor:
    """asteroid_filterbanks ParamSincFB.filters(): 40 cos + 40 sin band-pass filters (80, 1, 251)."""
    half = kernel_size // 2
    window_ = torch.from_numpy(np.hamming(kernel_size)[:half]).float()
    n_ = 2 * math.pi * (torch.arange(-half, 0.0).view(1, -1) / sample_rate)
    low = min_low_hz + torch.abs(low_hz_)
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), min_low_hz, sample_rate / 2)
    band = (high - low)[:, 0]
    ft_low, ft_high = torch.matmul(low, n_), torch.matmul(high, n_)
    out = []
    for kind in ("cos", "sin"):
        if kind == "cos":
            left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n_ / 2)) * window_
            center = 2 * band.view(-1, 1)
            right = torch.flip(left, dims=[1])
        else:
            left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n_ / 2)) * window_
            center = torch.zeros_like(band.view(-1, 1))
            right = -torch.flip(left, dims=[1])
        bp = torch.cat([left, center, right], dim=1) / (2 * band[:, None])
        out.append(bp.view(-1, 1, kernel_size))
    return torch.cat(out, dim=0)


def sincnet(sd: SD, wav: torch.Tensor) -> torch.Tensor:
    """pyannote.audio.models.blocks.sincnet.SincNet.forward; wav (B,1,S) -> (B,60,frames)."""
    x = F.instance_norm(wav, weight=sd["sincnet.wav_norm1d.weight"], bias=sd["sincnet.wav_norm1d.bias"], eps=1e-5)
    filt = sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])
    for c in range(3):
        if c == 0:
            x = torch.abs(F.conv1d(x, filt, stride=10))
        else:
            x = F.conv1d(x, sd[f"sincnet.conv1d.{c}.weight"], sd[f"sincnet.conv1d.{c}.bias"])
        x = F.max_pool1d(x, 3, stride=3)
        x = F.instance_norm(x, weight=sd[f"sincnet.norm1d.{c}.weight"], bias=sd[f"sincnet.norm1d.{c}.bias"], eps=1e-5)
        x = F.leaky_relu(x)
    return x


def lstm_layer(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> torch.Tensor:
    """One direction of one nn.LSTM layer (gate order i, f, g, o); x (B,T,in) -> (B,T,H)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xp = F.linear(x, w_ih, b_ih + b_hh)
    h = torch.zeros(B, H); c = torch.zeros(B, H)
    out = torch.empty(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = xp[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def pyannet(sd: SD, wav: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """PyanNet.forward: (B,1,160000) -> log-probabilities (B,589,7) over the powerset classes."""
    x = sincnet(sd, wav).transpose(1, 2)                 # (B, frames, 60)
    if taps is not None:
        taps["sincnet"] = x.clone()
    layer = 0
    while f"lstm.weight_ih_l{layer}" in sd:
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_layer(x, sd[f"lstm.weight_ih_l{layer}{suf}"], sd[f"lstm.weight_hh_l{layer}{suf}"],
                                   sd[f"lstm.bias_ih_l{layer}{suf}"], sd[f"lstm.bias_hh_l{layer}{suf}"], rev))
        x = torch.cat(outs, dim=-1)
        layer += 1
    if taps is not None:
        taps["lstm"] = x.clone()
    i = 0
    while f"linear.{i}.weight" in sd:
        x = F.leaky_relu(F.linear(x, sd[f"linear.{i}.weight"], sd[f"linear.{i}.bias"]))
        i += 1
    return F.log_softmax(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]), dim=-1)


def powerset_to_multilabel(logp: torch.Tensor) -> np.ndarray:
    """pyannote.audio.utils.powerset.Powerset.to_multilabel (hard): argmax class -> (.., 3) 0/1."""
    mapping = np.zeros((len(POWERSET), 3), np.float32)
    for k, s in enumerate(POWERSET):
        for spk in s:
            mapping[k, spk] = 1.0
    return mapping[logp.argmax(-1).numpy()]


# ------------------------------------------------------------------------------------ WeSpeaker ResNet34
def hamming_fbank(wave: np.ndarray) -> np.ndarray:
    """torchaudio.compliance.kaldi.fbank(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0,
    window_type='hamming', use_energy=False) on waveform * 32768 (pyannote WeSpeaker wrapper), then
    mean subtraction over time."""
    from . import fbank_ref as FB
    x = np.asarray(wave, np.float32) * np.float32(32768.0)
    win, shift, padded = 400, 160, 512
    m = FB.num_frames(x.shape[0], win, shift)
    idx = np.arange(m)[:, None] * shift + np.arange(win)[None, :]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    fr = fr - np.float32(0.97) * prev
    i = np.arange(win, dtype=np.float64)
    fr = fr * (0.54 - 0.46 * np.cos(2.0 * math.pi * i / (win - 1))).astype(np.float32)[None, :]
    spec = np.fft.rfft(fr.astype(np.float32), n=padded, axis=1)
    power = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2).astype(np.float32)
    banks = np.concatenate([FB.mel_banks(80, padded, 16000.0), np.zeros((80, 1), np.float32)], axis=1)
    feat = np.log(np.maximum(power @ banks.T, FB.EPS)).astype(np.float32)
    return feat - feat.mean(axis=0, keepdims=True)


def _bn(sd: SD, p: str, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def resnet34_trunk(sd: SD, feats: torch.Tensor) -> torch.Tensor:
    """wespeaker ResNet (BasicBlock [3,4,6,3], m_channels 32): feats (B,T,80) -> (B,256,10,T/8)."""
    x = feats.permute(0, 2, 1).unsqueeze(1)
    x = F.relu(_bn(sd, "resnet.bn1", F.conv2d(x, sd["resnet.conv1.weight"], padding=1)))
    for li, (nblk, stride) in enumerate(zip((3, 4, 6, 3), (1, 2, 2, 2)), start=1):
        for b in range(nblk):
            p = f"resnet.layer{li}.{b}"
            s = stride if b == 0 else 1
            out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], stride=s, padding=1)))
            out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1))
            sc = x
            if (p + ".shortcut.0.weight") in sd:
                sc = _bn(sd, p + ".shortcut.1", F.conv2d(x, sd[p + ".shortcut.0.weight"], stride=s))
            x = F.relu(out + sc)
    return x


def tstp(x: torch.Tensor, weights: Optional[torch.Tensor]) -> torch.Tensor:
    """pyannote.audio.models.blocks.pooling.StatsPool behind wespeaker's TSTP: x (B,C,F,T) -> (B, 2*C*F),
    (C,F) flattened channel-major, frame weights resampled to T by nearest interpolation."""
    B = x.shape[0]
    x = x.reshape(B, -1, x.shape[-1])
    if weights is None:
        return torch.cat([x.mean(-1), x.std(-1, unbiased=True)], dim=-1)
    w = F.interpolate(weights.unsqueeze(1), size=x.shape[-1], mode="nearest")        # (B,1,T')
    v1 = w.sum(-1) + 1e-8
    mean = (x * w).sum(-1) / v1
    dx2 = (x - mean.unsqueeze(-1)) ** 2
    v2 = (w * w).sum(-1)
    var = (dx2 * w).sum(-1) / (v1 - v2 / v1 + 1e-8)
    return torch.cat([mean, torch.sqrt(var)], dim=-1)


def wespeaker_embed(sd: SD, feats: torch.Tensor, weights: Optional[torch.Tensor]) -> torch.Tensor:
    stats = tstp(resnet34_trunk(sd, feats), weights)
    return F.linear(stats, sd["resnet.seg_1.weight"], sd["resnet.seg_1.bias"])
