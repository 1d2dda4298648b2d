"""Device resampler (reference front end: torchaudio.transforms.Resample at asr/wenet/cli/reverb.py:128-134) against
the numpy restatement of torchaudio's algorithm (oracle/resample_ref.py; parity unpinned, torchaudio is absent)."""
import math
import os

import numpy as np
import pytest

from oracle import resample_ref as RR
from reverb_amd import synth


def test_oracle_kernel_shapes_and_dc_gain():
    for orig, new in ((48000, 16000), (44100, 16000), (8000, 16000), (22050, 16000)):
        ker, width, o, n = RR.sinc_kernel(orig, new)
        assert ker.shape == (n, 2 * width + o)
        # every polyphase branch of a low-pass interpolator has (close to) unit DC gain
        assert np.allclose(ker.sum(axis=1), 1.0, atol=2e-2), (orig, new, ker.sum(axis=1))
    assert RR.sinc_kernel(48000, 16000)[1] == math.ceil(6 * 3 / 0.99)


def test_oracle_preserves_a_tone_and_the_length_rule():
    sr = 44100
    t = np.arange(int(0.5 * sr)) / sr
    x = 8000.0 * np.sin(2 * np.pi * 1000.0 * t)
    y = RR.resample(x, sr, 16000)
    assert y.shape[0] == math.ceil(160 * x.shape[0] / 441)
    ty = np.arange(y.shape[0]) / 16000.0
    want = 8000.0 * np.sin(2 * np.pi * 1000.0 * ty)
    assert np.abs(y[200:-200] - want[200:-200]).max() < 8000.0 * 5e-3
    # content above the new Nyquist is removed
    hi = RR.resample(8000.0 * np.sin(2 * np.pi * 12000.0 * t), sr, 16000)
    assert np.abs(hi[200:-200]).max() < 8000.0 * 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [48000, 44100, 8000, 22050])
def test_device_resampler_matches_oracle(rate):
    from reverb_amd.engine import Engine
    cfg = synth.make_config("tiny")
    eng = Engine(cfg, synth.make_state_dict(cfg, 0), dtype="f32", device=0, max_chunks=2, chunk_frames=400)
    rng = np.random.default_rng(rate)
    n = int(1.37 * rate) + 3
    t = np.arange(n) / rate
    x = (6000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * 3100 * t) + 500 * rng.standard_normal(n)).astype(np.int16)
    eng.upload_pcm(x, rate)
    got = eng.waveform()
    want = RR.resample(x.astype(np.float32), rate, 16000)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-2                 # int16 scale: fp32 summation order only
    # the fbank reads the un-rounded resampled float waveform
    from oracle import fbank_ref
    nf, feats = eng.fbank(return_feats=True)
    ref = fbank_ref.fbank(want)
    assert feats.shape == ref.shape
    assert np.abs(feats - ref).max() < 5e-3
    eng.close()


@pytest.mark.gpu
def test_transcribe_accepts_48k_wav(tmp_path):
    from reverb_amd.reverb import load_model
    mdir = synth.write_model_dir(str(tmp_path / "m"), "tiny")
    asr = load_model(mdir, gpu=0, dtype="f32", max_chunks=4)
    pcm16 = synth.synth_audio(6.0, seed=3)
    # a 48 kHz file carrying the same signal (3x oversampled by linear interpolation is enough for a smoke check)
    x48 = np.interp(np.arange(len(pcm16) * 3) / 3.0, np.arange(len(pcm16)), pcm16.astype(np.float64)).astype(np.int16)
    w48 = str(tmp_path / "a48.wav")
    synth.write_wav(w48, x48, sample_rate=48000)
    out = asr.transcribe(w48, mode="ctc_greedy_search", format="txt")
    assert isinstance(out, str)
    feats = asr.compute_feats(w48, num_mel_bins=80)
    assert feats.shape[1] == 1 + (math.ceil(len(x48) / 3) - 400) // 160
