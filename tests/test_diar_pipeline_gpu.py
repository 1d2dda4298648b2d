"""End-to-end diarization through the pyannote-shaped API (reference call sequence
diarization/infer_pyannote3.0.py:33-42) on synthetic weights: runs, is deterministic, writes well-formed
RTTM whose per-frame speaker count equals the pipeline's own instantaneous count."""
import io
import re

import numpy as np
import pytest

from reverb_amd import diarization as D
from reverb_amd import synth, synth_diar

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    d = tmp_path_factory.mktemp("diar")
    model = synth_diar.write_pipeline_dir(str(d / "pipe"))
    pcm = synth_diar.synth_conversation(47.3, seed=3)
    wav = str(d / "talk.wav")
    synth.write_wav(wav, pcm)
    return model, wav, pcm


def test_pipeline_runs_and_writes_rttm(setup):
    model, wav, pcm = setup
    pipe = D.Pipeline.from_pretrained(model, dtype="f32")
    pipe.to("cuda")
    ann = pipe(wav)
    assert ann.uri == "talk"
    buf = io.StringIO(); ann.write_rttm(buf)
    lines = buf.getvalue().splitlines()
    assert lines, "no speech found by a random-weight model is possible but not with this seed"
    pat = re.compile(r"^SPEAKER talk 1 \d+\.\d{3} \d+\.\d{3} <NA> <NA> SPEAKER_\d\d <NA> <NA>$")
    assert all(pat.match(l) for l in lines)
    starts = [float(l.split()[3]) for l in lines]
    assert starts == sorted(starts)
    assert max(float(l.split()[3]) + float(l.split()[4]) for l in lines) <= len(pcm) / 16000.0 + 10.0
    # same call again: identical output
    buf2 = io.StringIO(); pipe(wav).write_rttm(buf2)
    assert buf2.getvalue() == buf.getvalue()
    # waveform-dict input (pyannote's in-memory form) gives the same turns
    import torch
    ann3 = pipe({"waveform": torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None], "sample_rate": 16000, "uri": "talk"})
    buf3 = io.StringIO(); ann3.write_rttm(buf3)
    assert buf3.getvalue() == buf.getvalue()
    t = pipe.timings
    assert t["windows"] == 39 and t["embeddings"] <= 3 * 39


def test_pipeline_speaker_budget_and_count_consistency(setup):
    model, wav, pcm = setup
    pipe = D.Pipeline.from_pretrained(model, dtype="bf16").to("cuda")
    eng = pipe.engine
    W = eng.upload(pcm)
    binarized = D.powerset_to_multilabel(eng.segment())
    count = D.speaker_count(binarized, 1.0, 10.0)
    ann = pipe(wav, num_speakers=2)
    labels = ann.labels()
    assert 1 <= len(labels) <= 2
    # frame-level check: number of simultaneously active speakers in the RTTM == min(count, #clusters)
    n = count.shape[0]
    ts = np.arange(n) * D.FRAME_STEP + 0.5 * D.FRAME_DURATION
    active = np.zeros(n, int)
    for seg, _, _ in ann.itertracks(yield_label=True):
        active += ((ts >= seg.start - 1e-9) & (ts < seg.end - 1e-9))
    want = np.minimum(count[:, 0], len(labels))
    assert np.mean(active == want) > 0.97      # boundaries: a turn ends at the middle of the first inactive frame


def test_cli_script(setup, tmp_path):
    model, wav, _ = setup
    from reverb_amd.bin import infer_pyannote3
    infer_pyannote3.main([wav, "--out-dir", str(tmp_path / "out"), "--pipeline-model", model])
    text = (tmp_path / "out" / "talk.rttm").read_text()
    assert text.startswith("SPEAKER talk 1 ")


def test_joint_transcribe_diarize_writes_ctm_rttm_stm(setup, tmp_path):
    """BASELINE config 5 shape on one GPU: ASR CTM + diarization RTTM + word->speaker STM for one file."""
    model, wav, pcm = setup
    from reverb_amd.bin import transcribe_diarize
    asr_dir = synth.write_model_dir(str(tmp_path / "asr"), "tiny")
    transcribe_diarize.main([wav, "--asr-model", asr_dir, "--pipeline-model", model, "--out-dir", str(tmp_path / "o"), "--dtype", "f32"])
    ctm = (tmp_path / "o" / "talk.ctm").read_text().splitlines()
    rttm = (tmp_path / "o" / "talk.rttm").read_text().splitlines()
    stm = (tmp_path / "o" / "talk.stm").read_text().splitlines()
    assert len(ctm) == len(stm) > 0 and len(rttm) > 0
    speakers = {l.split()[7] for l in rttm}
    for c, s in zip(ctm, stm):
        cp, sp = c.split(" "), s.split(" ")
        assert sp[0] == "talk" and sp[1] == "1" and sp[2] in speakers
        assert abs(float(sp[3]) - float(cp[2])) < 1e-3 and sp[5] == cp[4]


def test_pipeline_reads_other_rates_and_flac(setup, tmp_path):
    """pyannote's Audio resamples to the model's rate and downmixes: a 48 kHz stereo WAV and a 16 kHz FLAC go through the same
    pipeline.  The FLAC holds the very samples of the WAV fixture (identical RTTM); the device resampler is checked against
    the torchaudio restatement (oracle/resample_ref.py) up to the int16 rounding."""
    from oracle import resample_ref
    from tests import flac_writer as FW
    from tests.test_audio_decode import wav_bytes
    model, wav, pcm = setup
    pipe = D.Pipeline.from_pretrained(model, dtype="f32").to("cuda")
    want = io.StringIO(); pipe(wav).write_rttm(want)
    flac = tmp_path / "talk.flac"
    flac.write_bytes(FW.encode(pcm.astype(np.int64)[None], 16, 16000, block=4096, predictor=("fixed", 2), partition_order=3))
    got = io.StringIO(); pipe(str(flac)).write_rttm(got)
    assert got.getvalue() == want.getvalue()
    # 48 kHz stereo: both channels carry the signal 3x oversampled (linear interpolation is enough here)
    x48 = np.interp(np.arange(len(pcm) * 3) / 3.0, np.arange(len(pcm)), pcm.astype(np.float64)).astype(np.int16)
    w48 = tmp_path / "talk48.wav"
    w48.write_bytes(wav_bytes(1, 2, 48000, 16, np.stack([x48, x48], axis=1).astype("<i2").tobytes()))
    loaded, uri = pipe._load(str(w48))
    ref = resample_ref.resample(x48.astype(np.float32), 48000, 16000)
    assert uri == "talk48" and loaded.dtype == np.int16 and loaded.shape == ref.shape
    assert np.abs(loaded.astype(np.float32) - ref).max() <= 0.5 + 2e-2         # half an LSB of rounding + fp32 summation order
    ann = pipe(str(w48))
    assert ann.uri == "talk48" and len(ann.labels()) >= 1
    # the in-memory form at another rate
    import torch
    ann2 = pipe({"waveform": torch.from_numpy(x48.astype(np.float32) / 32768.0)[None], "sample_rate": 48000, "uri": "talk48"})
    a, b = io.StringIO(), io.StringIO()
    ann.write_rttm(a); ann2.write_rttm(b)
    assert a.getvalue() == b.getvalue()
