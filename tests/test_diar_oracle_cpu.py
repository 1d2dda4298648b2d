"""Self-consistency of the diarization oracle (oracle/diar_ref.py).  pyannote.audio is not available, so these are not
parity pins (the oracle's header says PARITY UNPINNED); they check the restatement against the architecture facts
the reference relies on: frame counts, receptive field, pooling identities, parameter counts of the published models."""
import numpy as np
import torch

from oracle import diar_ref as R
from reverb_amd import synth_diar as SD


def test_sincnet_frame_geometry():
    """160000 samples -> 15975 -> 5325 -> 5321 -> 1773 -> 1769 -> 589 frames; receptive field 991, step 270 samples."""
    cfg = SD.make_diar_config()
    sd = R.to_torch_sd(SD.make_segmentation_sd(cfg, 0))
    with torch.no_grad():
        assert R.sincnet(sd, torch.zeros(1, 1, 160000)).shape == (1, 60, R.NUM_FRAMES)
        # frame k needs samples up to 991 + 270 k (instance norm refuses a single frame, so start at two)
        assert R.sincnet(sd, torch.zeros(1, 1, R.FRAME_SIZE + R.FRAME_STEP)).shape[-1] == 2
        assert R.sincnet(sd, torch.zeros(1, 1, R.FRAME_SIZE + 2 * R.FRAME_STEP - 1)).shape[-1] == 2
        assert R.sincnet(sd, torch.zeros(1, 1, R.FRAME_SIZE + 2 * R.FRAME_STEP)).shape[-1] == 3


def test_sinc_filters_are_symmetric_cos_and_antisymmetric_sin_bandpasses():
    cfg = SD.make_diar_config()
    sd = R.to_torch_sd(SD.make_segmentation_sd(cfg, 0))
    f = R.sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])[:, 0]
    assert f.shape == (80, 251)
    assert torch.allclose(f[:40], f[:40].flip(1)) and torch.allclose(f[40:], -f[40:].flip(1))
    assert torch.all(f[40:, 125] == 0) and torch.allclose(f[:40, 125], torch.ones(40))
    # a cosine filter passes a tone inside its band and rejects one far outside
    low = 50 + sd["sincnet.conv1d.0.filterbank.low_hz_"][10, 0].abs()
    band = 50 + sd["sincnet.conv1d.0.filterbank.band_hz_"][10, 0].abs()
    t = torch.arange(251) / 16000.0
    inside = (f[10] * torch.cos(2 * np.pi * (low + band / 2) * t)).sum().abs()
    outside = (f[10] * torch.cos(2 * np.pi * (low + band * 6 + 500) * t)).sum().abs()
    assert inside > 5 * outside


def test_parameter_counts_match_the_published_models():
    cfg = SD.make_diar_config()
    seg = SD.make_segmentation_sd(cfg, 0)
    emb = SD.make_embedding_sd(cfg, 0)
    n_seg = sum(v.size for v in seg.values())
    n_emb = sum(v.size for k, v in emb.items() if "running" not in k)
    assert 1.4e6 < n_seg < 1.6e6            # pyannote/segmentation-3.0: ~1.5 M parameters
    assert 6.5e6 < n_emb < 6.8e6            # wespeaker ResNet34 (embed 256): ~6.6 M parameters


def test_stats_pooling_identities():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((2, 4, 3, 50)).astype(np.float32))
    ones = torch.ones(2, 589)
    a, b = R.tstp(x, ones), R.tstp(x, None)
    assert torch.allclose(a, b, atol=1e-5)                                  # all-ones weights = plain mean / unbiased std
    w = torch.zeros(2, 589); w[:, :295] = 1.0                               # first half of the frames ...
    half = R.tstp(x, w)
    T = x.shape[-1]
    keep = [t for t in range(T) if int(np.floor(t * np.float32(589 / T))) < 295]     # ... = these trunk frames (nearest)
    ref = R.tstp(x[..., keep], None)
    assert torch.allclose(half, ref, atol=1e-5)
    zero = R.tstp(x, torch.zeros(2, 589))
    assert torch.all(zero == 0)                                             # an inactive speaker pools to zeros


def test_powerset_decoding_and_resnet_shapes():
    logp = torch.log_softmax(torch.randn(3, 10, 7), -1)
    ml = R.powerset_to_multilabel(logp)
    assert ml.shape == (3, 10, 3) and ml.sum(-1).max() <= 2
    cfg = SD.make_diar_config()
    sd = R.to_torch_sd(SD.make_embedding_sd(cfg, 0))
    with torch.no_grad():
        out = R.resnet34_trunk(sd, torch.zeros(1, 998, 80))
        assert out.shape == (1, 256, 10, 125)
        e = R.wespeaker_embed(sd, torch.zeros(1, 998, 80), torch.ones(1, 589))
        assert e.shape == (1, 256)
    assert R.hamming_fbank(np.zeros(160000, np.float32)).shape == (998, 80)
