"""TEST INFRASTRUCTURE.  Pins for the two oracles that are UNPINNED in this repository because the third-party packages
the reference calls are not installable here (no network): run this ONCE on any machine that has them and commit the
files it writes under tests/golden/ -- tests/test_thirdparty_fixtures.py and the `-m gpu` fixture tests then check
`oracle/fbank_ref.py`, `oracle/resample_ref.py`, `oracle/diar_ref.py` AND the HIP kernels against the real packages.

    pip install torch==2.2.2 torchaudio==2.2.2            # asr/requirements.txt:1-2
    python -m oracle.make_thirdparty_fixtures torchaudio
    pip install pyannote.audio==3.3.1                     # diarization/requirements.txt:1
    python -m oracle.make_thirdparty_fixtures pyannote

NOT RUN in the build container (packages absent): the call signatures below follow the reference's call sites
(asr/wenet/cli/reverb.py:128-144, diarization/infer_pyannote3.0.py:33-42) and the packages' documented APIs.
Inputs are regenerated from seeds by reverb_amd.synth / synth_diar, so only outputs are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

FBANK_AUDIO = dict(seconds=3.0, seed=99)                      # synth.synth_audio(**FBANK_AUDIO)
RESAMPLE_RATES = (8000, 22050, 44100, 48000)                  # synth.synth_audio(1.0, seed=7, sample_rate=r)
DIAR_AUDIO = dict(seconds=14.3, seed=11)                      # synth_diar.synth_conversation(**DIAR_AUDIO)


def torchaudio_fixtures():
    import torch
    import torchaudio
    import torchaudio.compliance.kaldi as kaldi
    from reverb_amd import synth
    pcm = synth.synth_audio(**FBANK_AUDIO)
    wave = torch.from_numpy(pcm.astype(np.float32)).unsqueeze(0)          # torchaudio.load(normalize=False).to(float): int16 scale
    feats = kaldi.fbank(wave, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0, energy_floor=0.0,
                        sample_frequency=16000)                           # cli/reverb.py:136-144
    out = {"fbank": feats.numpy().astype(np.float32), "torchaudio_version": np.array(torchaudio.__version__)}
    for r in RESAMPLE_RATES:
        x = synth.synth_audio(1.0, seed=7, sample_rate=r)
        y = torchaudio.transforms.Resample(r, 16000)(torch.from_numpy(x.astype(np.float32)).unsqueeze(0))   # cli/reverb.py:131-134
        out[f"resample_{r}"] = y[0].numpy().astype(np.float32)
    out.update(audio_reader_fixtures(torch, torchaudio))
    np.savez_compressed(os.path.join(GOLDEN, "thirdparty_torchaudio.npz"), **out)
    print("wrote thirdparty_torchaudio.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


AUDIO_FILES = (("wav_s16", "wav", dict(encoding="PCM_S", bits_per_sample=16)), ("wav_u8", "wav", dict(encoding="PCM_U", bits_per_sample=8)),
               ("wav_s24", "wav", dict(encoding="PCM_S", bits_per_sample=24)), ("wav_s32", "wav", dict(encoding="PCM_S", bits_per_sample=32)),
               ("wav_f32", "wav", dict(encoding="PCM_F", bits_per_sample=32)), ("wav_ulaw", "wav", dict(encoding="ULAW", bits_per_sample=8)),
               ("wav_alaw", "wav", dict(encoding="ALAW", bits_per_sample=8)), ("flac_16", "flac", dict(bits_per_sample=16)),
               ("flac_24", "flac", dict(bits_per_sample=24)))


def audio_reader_fixtures(torch, torchaudio):
    """The VALUE convention of `torchaudio.load(path, normalize=False)` per container (cli/reverb.py:128): every file below is
    written with torchaudio.save from one seeded stereo signal, its BYTES are stored, and so is what load(normalize=False)
    returns for it (dtype name + values) -- tests/test_thirdparty_fixtures.py decodes the stored bytes with librvb's reader
    (reverb_amd/audio.py) and compares."""
    import tempfile
    from reverb_amd import synth
    x = synth.synth_audio(0.5, seed=21).astype(np.float32) / 32768.0
    wave = torch.from_numpy(np.stack([x, -0.5 * x]))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, ext, kw in AUDIO_FILES:
            path = os.path.join(d, name + "." + ext)
            torchaudio.save(path, wave, 16000, **kw)
            got, rate = torchaudio.load(path, normalize=False)
            assert rate == 16000
            with open(path, "rb") as f:
                out["audio_bytes_" + name] = np.frombuffer(f.read(), np.uint8)
            out["audio_dtype_" + name] = np.array(str(got.dtype).replace("torch.", ""))
            out["audio_float_" + name] = got.to(torch.float).numpy()          # cli/reverb.py:130
    return out


def pyannote_fixtures():
    import torch
    import pyannote.audio
    from pyannote.audio.core.task import Problem, Resolution, Specifications
    from pyannote.audio.models.embedding import WeSpeakerResNet34
    from pyannote.audio.models.segmentation import PyanNet
    from reverb_amd import synth_diar as SD
    cfg = SD.make_diar_config()
    seg_sd, emb_sd = SD.make_segmentation_sd(cfg, 0), SD.make_embedding_sd(cfg, 0)
    pcm = SD.synth_conversation(**DIAR_AUDIO)
    wav = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
    n_win = (len(wav) - 160000) // 16000 + 1
    x = torch.stack([wav[w * 16000:w * 16000 + 160000] for w in range(n_win)])[:, None]      # 10 s windows every 1 s

    seg = PyanNet(sample_rate=16000, num_channels=1, sincnet={"stride": 10},
                  lstm={"hidden_size": 128, "num_layers": 4, "bidirectional": True, "monolithic": True, "dropout": 0.0},
                  linear={"hidden_size": 128, "num_layers": 2})
    seg.specifications = Specifications(problem=Problem.MONO_LABEL_CLASSIFICATION, resolution=Resolution.FRAME, duration=10.0,
                                        classes=["1", "2", "3"], powerset_max_classes=2)        # pyannote/segmentation-3.0
    seg.build()
    seg.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in seg_sd.items()}, strict=True)
    seg.eval()
    emb = WeSpeakerResNet34()
    missing = emb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in emb_sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    emb.eval()
    with torch.no_grad():
        logp = seg(x)                                                     # (W, 589, 7) log-softmax over the powerset
        masks = torch.zeros(n_win, 589)
        masks[:, 50:400] = 1.0
        e = emb(x, weights=masks)                                         # (W, 256)
    np.savez_compressed(os.path.join(GOLDEN, "thirdparty_pyannote.npz"), seg_logp=logp.numpy().astype(np.float32),
                        emb=e.numpy().astype(np.float32), pyannote_version=np.array(pyannote.audio.__version__))
    print("wrote thirdparty_pyannote.npz", logp.shape, e.shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["torchaudio", "pyannote"]
    os.makedirs(GOLDEN, exist_ok=True)
    if "torchaudio" in which:
        torchaudio_fixtures()
    if "pyannote" in which:
        pyannote_fixtures()
