"""Parity of the HIP diarization networks (rvd_* C ABI) with the oracle's restatement of the
pyannote architectures (oracle/diar_ref.py; parity unpinned -- pyannote.audio is not available
here, see the oracle's header) on seeded synthetic weights."""
import numpy as np
import pytest
import torch

from conftest import HAVE_GPU
from oracle import diar_ref as R
from reverb_amd import synth_diar as SD

pytestmark = pytest.mark.gpu


def windows_of(pcm, cfg):
    wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
    n, win, step = wav.shape[0], cfg["window_samples"], cfg["step_samples"]
    full = (n - win) // step + 1 if n >= win else 0
    tail = n < win or (n - win) % step > 0
    W = full + (1 if tail else 0)
    x = torch.zeros(W, 1, win)
    for w in range(W):
        seg = wav[w * step:w * step + win]
        x[w, 0, :seg.shape[0]] = seg
    return x


@pytest.fixture(scope="module")
def case():
    cfg = SD.make_diar_config()
    seg_sd = SD.make_segmentation_sd(cfg, 0)
    pcm = SD.synth_conversation(14.3, seed=11)
    x = windows_of(pcm, cfg)
    taps = {}
    with torch.no_grad():
        logp = R.pyannet(R.to_torch_sd(seg_sd), x, taps)
    return dict(cfg=cfg, seg_sd=seg_sd, pcm=pcm, x=x, logp=logp.numpy(), taps={k: v.numpy() for k, v in taps.items()})


def test_window_count_matches_pyannote_slide(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    assert eng.frames == R.NUM_FRAMES
    for n, want in ((1, 1), (159999, 1), (160000, 1), (160001, 2), (176000, 2), (176001, 3), (228800, 6), (57600000, 3591)):
        assert eng.num_windows(n) == want, n
    eng.close()


def test_segmentation_f32_matches_oracle(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    W = eng.upload(case["pcm"])
    assert W == case["x"].shape[0] == 6
    logp = eng.segment()
    sinc = eng.tap("sincnet", W)
    lstm = eng.tap("lstm", W)
    # fp32 everywhere; differences are summation order (sinc bank shared across windows, MFMA K order)
    assert np.abs(sinc - case["taps"]["sincnet"]).max() < 2e-3
    assert np.abs(lstm - case["taps"]["lstm"]).max() < 2e-3
    assert np.abs(logp - case["logp"]).max() < 1e-2
    agree = (logp.argmax(-1) == case["logp"].argmax(-1)).mean()
    assert agree > 0.995, agree
    assert np.allclose(np.exp(logp).sum(-1), 1.0, atol=1e-4)
    eng.close()


def test_segmentation_f32_batching_is_invariant(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    eng.upload(case["pcm"])
    a = eng.segment(batch=512)
    b = eng.segment(batch=4)            # 4 + 2 windows
    c = eng.segment(first=3, n=2)
    assert np.array_equal(a, b)
    assert np.array_equal(a[3:5], c)
    eng.close()


def test_segmentation_bf16_close_to_oracle(case):
    from reverb_amd.diar_engine import DiarEngine
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="bf16")
    W = eng.upload(case["pcm"])
    logp = eng.segment()
    sinc = eng.tap("sincnet", W)
    ref = case["taps"]["sincnet"]
    assert np.abs(sinc - ref).mean() < 0.02 * np.abs(ref).mean() + 1e-3
    # bf16 inputs through 4 recurrent layers of 589 steps: compare distributions, not bits
    p, q = np.exp(logp), np.exp(case["logp"])
    assert np.abs(p - q).mean() < 0.02
    agree = (logp.argmax(-1) == case["logp"].argmax(-1)).mean()
    assert agree > 0.93, agree
    eng.close()


def test_short_audio_single_padded_window(case):
    from reverb_amd.diar_engine import DiarEngine
    pcm = case["pcm"][:52345]
    x = windows_of(pcm, case["cfg"])
    with torch.no_grad():
        want = R.pyannet(R.to_torch_sd(case["seg_sd"]), x).numpy()
    eng = DiarEngine(case["cfg"], case["seg_sd"], dtype="f32")
    assert eng.upload(pcm) == 1
    got = eng.segment()
    assert np.abs(got - want).max() < 1e-2
    eng.close()
