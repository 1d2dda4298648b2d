"""TEST INFRASTRUCTURE ONLY (oracle).  PARITY UNPINNED: the resampler of the reference front end is
`torchaudio.transforms.Resample(sample_rate, 16000)` (asr/wenet/cli/reverb.py:128-134), a third-party function
(torchaudio==2.2.2, asr/requirements.txt:1) that is neither vendored nor installed, and the reference has no test
for it.  This is a numpy restatement of torchaudio.functional.resample's published algorithm
(`_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`: sinc_interp_hann, lowpass_filter_width=6,
rolloff=0.99, kernel built in float64 and applied in float32)."""
import math

import numpy as np


def sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels = kernels * window * scale
    return kernels.astype(np.float32), width, orig, new


def resample(wave: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """wave: 1-D float array -> float32 resampled array of length ceil(new * len / orig)."""
    if orig_freq == new_freq:
        return np.asarray(wave, np.float32)
    ker, width, orig, new = sinc_kernel(orig_freq, new_freq)
    x = np.asarray(wave, np.float32)
    n = x.shape[0]
    xp = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + orig, np.float32)])
    K = ker.shape[1]
    nfr = (xp.shape[0] - K) // orig + 1
    frames = np.lib.stride_tricks.as_strided(xp, (nfr, K), (xp.strides[0] * orig, xp.strides[0]))
    out = (frames @ ker.T).astype(np.float32).reshape(-1)            # (frames, new) -> interleaved
    return out[: math.ceil(new * n / orig)]
