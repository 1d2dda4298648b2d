"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

CPU restatement (numpy, float32 arithmetic where Kaldi uses float) of the log-mel
filterbank front end the reference calls at `asr/wenet/cli/reverb.py:136-144`:

    kaldi.fbank(waveform, num_mel_bins=80, frame_length=25, frame_shift=10,
                dither=0.0, energy_floor=0.0, sample_frequency=16000)

PARITY UNPINNED: the algorithm lives in the third-party dependency
`torchaudio==2.2.2` (`asr/requirements.txt:1`), function
`torchaudio.compliance.kaldi.fbank`, which is neither vendored under
/root/reference nor installed here, and the reference has no test that pins
it.  This file restates the published Kaldi-compatible algorithm
(torchaudio 2.2.2 compliance/kaldi.py defaults: snip_edges=True,
remove_dc_offset=True, preemphasis 0.97, povey window, round_to_power_of_two,
low_freq 20, high_freq 0 (=Nyquist), use_power=True, use_log_fbank=True,
use_energy=False, subtract_mean=False, channel 0).  The independent numpy
implementation in `transformers.audio_utils` is used as a cross-check in
tests/test_fbank_oracle.py.
"""
import math

import numpy as np

EPS = np.float32(1.1920928955078125e-07)  # torch.finfo(torch.float32).eps


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins=80, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """(num_bins, padded//2) triangular filters in mel space (Kaldi get_mel_banks,
    no VTLN).  The caller right-pads one zero column for the Nyquist bin."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel_low = mel_scale(low_freq)
    mel_high = mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(0.0, np.minimum(up, down)).astype(np.float32)


def povey_window(n=400):
    # hann(periodic=False) ** 0.85
    i = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * math.pi * i / (n - 1))) ** 0.85).astype(np.float32)


def num_frames(num_samples, win=400, shift=160):
    if num_samples < win:
        return 0
    return 1 + (num_samples - win) // shift


def fbank(waveform, num_mel_bins=80, frame_length_ms=25, frame_shift_ms=10, sample_freq=16000,
          preemph=0.97):
    """waveform: 1-D array (int16 or float) in int16 scale (reference loads with
    normalize=False, reverb.py:128).  Returns (m, num_mel_bins) float32."""
    x = np.asarray(waveform).astype(np.float32)
    win = int(sample_freq * frame_length_ms * 0.001)
    shift = int(sample_freq * frame_shift_ms * 0.001)
    padded = 1
    while padded < win:
        padded *= 2
    m = num_frames(x.shape[0], win, shift)
    if m == 0:
        return np.zeros((0, num_mel_bins), np.float32)
    idx = np.arange(m)[:, None] * shift + np.arange(win)[None, :]
    frames = x[idx]                                       # (m, win) strided view copy
    frames = frames - frames.mean(axis=1, keepdims=True, dtype=np.float32)   # remove_dc_offset
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)           # replicate pad
    frames = frames - np.float32(preemph) * prev
    frames = frames * povey_window(win)[None, :]
    spec = np.fft.rfft(frames.astype(np.float32), n=padded, axis=1)
    power = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2).astype(np.float32)
    banks = mel_banks(num_mel_bins, padded, float(sample_freq))              # (bins, padded/2)
    banks = np.concatenate([banks, np.zeros((num_mel_bins, 1), np.float32)], axis=1)
    mel = power @ banks.T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)
