"""The host audio reader (csrc/audio.cpp behind `rvb_audio_*`, reverb_amd/audio.py) -- the stand-in for
`torchaudio.load(audio_file, normalize=False)` of asr/wenet/cli/reverb.py:128 -- on RIFF/WAVE and FLAC.  No GPU is needed:
the decoder is host code in librvb.

FLAC known answers: the complete example file of RFC 9639 appendix D.1 (the decoder verifies its CRC-8, CRC-16 and the
MD5 signature of the decoded PCM), then streams from tests/flac_writer.py covering every subframe kind, residual coding,
stereo decorrelation, sample size and header variant, each carrying hashlib's MD5 of the PCM."""
import struct

import numpy as np
import pytest

from reverb_amd import audio
from tests import flac_writer as FW

RFC9639_D1 = bytes.fromhex("664c6143800000221000100000000f00000f0ac442f0000000013e84b41807dc690307586a3dad1a2e0f"
                           "fff869180000bf0358fd03128baa9a")


def signal(nch, n, bps, seed, kind="tones"):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    peak = (1 << (bps - 1)) - 1
    if kind == "noise":
        x = rng.integers(-peak - 1, peak + 1, size=(nch, n))
    else:
        base = 0.45 * np.sin(2 * np.pi * t / 57.0) + 0.3 * np.sin(2 * np.pi * t / 13.1 + 1.0)
        x = np.stack([np.round(peak * (base * (1 - 0.2 * c) + 0.02 * rng.standard_normal(n) + 0.05 * c)) for c in range(nch)])
    return np.clip(x, -peak - 1, peak).astype(np.int64)


def native(x, bps):
    """what the decoder returns: int16 / float32(int32), left-justified"""
    if bps <= 16:
        return (x << (16 - bps)).astype(np.int16)
    return (x << (32 - bps)).astype(np.int64).astype(np.float32)


def check_roundtrip(x, bps, rate=16000, **kw):
    data = FW.encode(x, bps, rate, **kw)
    got, info = audio.decode_bytes(data)
    assert (info.container, info.channels, info.sample_rate, info.bits_per_sample, info.frames) == \
        ("flac", x.shape[0], rate, bps, x.shape[1])
    assert info.sample_format == ("int16" if bps <= 16 else "int32")
    assert info.md5_checked == kw.get("md5", True)
    want = native(x, bps)
    assert got.dtype == want.dtype
    np.testing.assert_array_equal(got, want)
    return data


def test_rfc9639_example_file():
    got, info = audio.decode_bytes(RFC9639_D1)
    assert info == audio.AudioInfo("flac", "int16", 2, 44100, 16, 1, True)
    assert got.tolist() == [[25588], [10416]]
    assert audio.probe_bytes(RFC9639_D1) == info._replace(md5_checked=False)


@pytest.mark.parametrize("predictor", ["verbatim", ("fixed", 0), ("fixed", 1), ("fixed", 2), ("fixed", 3), ("fixed", 4),
                                       ("lpc", 1, 5), ("lpc", 8, 12), ("lpc", 12, 15), ("lpc", 32, 14), "auto"])
def test_flac_subframe_kinds(predictor):
    check_roundtrip(signal(1, 1500, 16, 1), 16, block=576, predictor=predictor)


@pytest.mark.parametrize("stereo", ["independent", "left_side", "side_right", "mid_side"])
@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24, 32])
def test_flac_stereo_modes_and_sample_sizes(stereo, bps):
    # full-scale noise in the last case of each size: the side channel needs bps + 1 bits (33 at 32 bits per sample)
    check_roundtrip(signal(2, 700, bps, bps, "tones"), bps, rate=44100, block=256, stereo=stereo)
    check_roundtrip(signal(2, 64, bps, 100 + bps, "noise"), bps, rate=48000, block=64, stereo=stereo, predictor="verbatim")


def test_flac_constant_wasted_bits_and_silence():
    x = signal(2, 1024, 16, 3)
    x[0] = (x[0] >> 5) << 5                  # 5 wasted bits
    x[1, :512] = -77                          # one constant block
    x[1, 512:] = 0                            # digital silence
    check_roundtrip(x, 16, block=512)
    check_roundtrip(x, 16, block=512, stereo="mid_side")
    y = signal(1, 300, 24, 4) << 0
    y = (y >> 9) << 9
    check_roundtrip(y, 24, block=300, predictor=("fixed", 2))


def test_flac_residual_codings():
    x = signal(1, 4096, 16, 5)
    for po in (0, 1, 4, 8):
        check_roundtrip(x, 16, block=4096, predictor=("fixed", 2), partition_order=po)
    check_roundtrip(x, 16, block=4096, predictor=("lpc", 8, 12), escape_first=True)           # escape-coded partition
    check_roundtrip(x, 16, block=4096, predictor=("fixed", 1), force_rice2=True)             # 5-bit Rice parameters
    z = np.zeros((1, 512), np.int64)
    z[0, 200] = 1000                                                                       # escape with width 0 partitions around it
    check_roundtrip(z, 16, block=512, predictor=("fixed", 0), partition_order=3, escape_first=True)
    check_roundtrip(signal(1, 2048, 24, 6, "noise"), 24, block=2048, predictor=("fixed", 0))   # Rice parameter >= 15 -> method 1


@pytest.mark.parametrize("block", [16, 192, 255, 256, 1000, 1152, 4608, 65535])
def test_flac_block_size_codes(block):
    n = block * 2 + min(block - 1, 37)                       # two full blocks and a short last one
    x = signal(1, n, 16, block)
    check_roundtrip(x, 16, block=block, predictor=("fixed", 1) if block > 5000 else "auto")


@pytest.mark.parametrize("rate,code", [(16000, None), (16000, 0), (44100, None), (37000, 12), (11025, 13), (65534, 13), (352800, 14),
                                       (96000, None)])
def test_flac_sample_rate_codes(rate, code):
    check_roundtrip(signal(1, 400, 16, rate % 97), 16, rate=rate, block=192, rate_code=code, explicit_bps=code is None)


def test_flac_stream_variants():
    x = signal(2, 5000, 16, 7)
    check_roundtrip(x, 16, block=1024, id3=True)                                              # ID3v2 tag in front
    check_roundtrip(x, 16, block=1024, padding=False)                                         # STREAMINFO is the only block
    check_roundtrip(x, 16, block=1024, md5=False)                                             # no signature: nothing to check
    check_roundtrip(x, 16, block=1024, total_known=False)                                     # sample count unknown
    check_roundtrip(x, 16, variable_blocks=[1000, 16, 2500, 1484], stereo="mid_side")          # variable block size
    data = check_roundtrip(x, 16, block=1024) + b"TAG" + bytes(125)                            # ID3v1 trailer
    got, _ = audio.decode_bytes(data)
    np.testing.assert_array_equal(got, x.astype(np.int16))
    many = signal(1, 16 * 300, 16, 8)
    check_roundtrip(many, 16, block=16)                                                        # frame numbers with 2-byte codes
    one, info = audio.decode_bytes(FW.encode(x, 16, 16000, block=1024), channel=1)
    assert one.shape == (1, 5000)
    np.testing.assert_array_equal(one[0], x[1].astype(np.int16))


def test_flac_eight_channels():
    check_roundtrip(signal(8, 300, 16, 9), 16, block=256)


def test_flac_corruption_is_detected():
    x = signal(2, 3000, 16, 10)
    good = FW.encode(x, 16, 16000, block=1024, stereo="left_side")
    first = good.index(b"\xff\xf8", 42)
    for at, what in ((first + 2, "CRC-8"), (first + 40, "CRC-16|residual|subframe|LPC|wasted"), (len(good) - 1, "CRC-16")):
        bad = bytearray(good)
        bad[at] ^= 0x10
        with pytest.raises(ValueError, match=what):
            audio.decode_bytes(bytes(bad))
    bad = bytearray(good)
    bad[8 + 18] ^= 1                                          # a bit of the MD5 signature
    with pytest.raises(ValueError, match="MD5"):
        audio.decode_bytes(bytes(bad))
    with pytest.raises(ValueError, match="ends inside|truncated|declares"):
        audio.decode_bytes(good[:len(good) - 300])
    with pytest.raises(ValueError, match="STREAMINFO"):
        audio.decode_bytes(b"fLaC" + bytes([0x84, 0, 0, 4]) + bytes(4))
    # a frame whose MD5-less stream was altered consistently (CRCs recomputed) still decodes: the CRCs are what is checked
    assert audio.decode_bytes(FW.encode(x, 16, 16000, block=1024, md5=False))[0].shape == (2, 3000)


def test_flac_long_stream_matches():
    """30 s of 16 kHz mono speech-like audio, the front end's own input shape"""
    from reverb_amd import synth
    pcm = synth.synth_audio(30.0, seed=4).astype(np.int64)[None]
    check_roundtrip(pcm, 16, block=4096, predictor=("fixed", 2), partition_order=4)


# ----------------------------------------------------------------------------------------------------- RIFF / WAVE
def wav_bytes(tag, channels, rate, bits, payload, extensible=False, extra_chunks=b"", data_size=None):
    align = channels * bits // 8
    fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, channels, rate, rate * align, align, bits)
    if extensible:
        fmt += struct.pack("<HHIH", 22, bits, 3, tag) + bytes.fromhex("000000001000800000aa00389b71")
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra_chunks + b"data" + \
        struct.pack("<I", len(payload) if data_size is None else data_size) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_native_formats():
    rng = np.random.default_rng(0)
    n = 257
    s16 = rng.integers(-32768, 32768, size=(n, 2)).astype("<i2")
    got, info = audio.decode_bytes(wav_bytes(1, 2, 8000, 16, s16.tobytes()))
    assert info == audio.AudioInfo("wav", "int16", 2, 8000, 16, n, False) and got.dtype == np.int16
    np.testing.assert_array_equal(got, s16.T)
    got, info = audio.decode_bytes(wav_bytes(1, 2, 8000, 16, s16.tobytes(), extensible=True,
                                             extra_chunks=b"LIST" + struct.pack("<I", 5) + b"INFOx\x00"))   # odd chunk: pad byte
    np.testing.assert_array_equal(got, s16.T)
    u8 = rng.integers(0, 256, size=(n, 1)).astype(np.uint8)
    got, info = audio.decode_bytes(wav_bytes(1, 1, 16000, 8, u8.tobytes()))
    assert info.sample_format == "uint8" and got.dtype == np.float32
    np.testing.assert_array_equal(got, u8.T.astype(np.float32))
    s24 = rng.integers(-(1 << 23), 1 << 23, size=(n, 2))
    raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in s24.reshape(-1))
    got, info = audio.decode_bytes(wav_bytes(1, 2, 48000, 24, raw))
    assert info.sample_format == "int32"
    np.testing.assert_array_equal(got, (s24.T << 8).astype(np.float32))           # left-justified int32, then .to(float)
    s32 = rng.integers(-(1 << 31), 1 << 31, size=(n, 1)).astype("<i4")
    got, _ = audio.decode_bytes(wav_bytes(1, 1, 16000, 32, s32.tobytes()))
    np.testing.assert_array_equal(got, s32.T.astype(np.float32))
    f32 = rng.standard_normal((n, 2)).astype("<f4")
    got, info = audio.decode_bytes(wav_bytes(3, 2, 16000, 32, f32.tobytes()))
    assert info.sample_format == "float32"
    np.testing.assert_array_equal(got, f32.T)
    f64 = rng.standard_normal((n, 1)).astype("<f8")
    got, info = audio.decode_bytes(wav_bytes(3, 1, 16000, 64, f64.tobytes()))
    assert info.sample_format == "float64"
    np.testing.assert_array_equal(got, f64.T.astype(np.float32))


def test_wav_g711():
    codes = np.arange(256, dtype=np.uint8)
    a, info = audio.decode_bytes(wav_bytes(6, 1, 8000, 8, codes.tobytes()))
    u, _ = audio.decode_bytes(wav_bytes(7, 1, 8000, 8, codes.tobytes()))
    assert info.sample_format == "int16" and a.dtype == np.int16
    a, u = a[0].astype(int), u[0].astype(int)
    # G.711 anchor points
    assert (a[0xD5], a[0x55], a[0xAA], a[0x2A]) == (8, -8, 32256, -32256)
    assert (u[0xFF], u[0x7F], u[0x80], u[0x00]) == (0, 0, 32124, -32124)
    # both laws are odd-symmetric in the sign bit and strictly monotonic in the magnitude code after the bit inversions
    np.testing.assert_array_equal(a[codes ^ 0x80], -a)
    np.testing.assert_array_equal(u[codes ^ 0x80], -u)
    assert np.all(np.diff(a[(np.arange(128) ^ 0x55) | 0x80]) > 0)
    assert np.all(np.diff(u[0xFF - np.arange(128)]) > 0)
    # segment structure: 16 steps per chord, the step doubles from chord to chord
    mag = u[0xFF - np.arange(128)]
    steps = np.diff(mag).reshape(-1)
    assert set(steps[:15]) == {8} and set(steps[16:31]) == {16} and set(steps[112:127]) == {1024}


def test_wav_truncated_and_streamed_headers():
    s16 = np.arange(-50, 50, dtype="<i2")
    whole = wav_bytes(1, 1, 16000, 16, s16.tobytes())
    got, _ = audio.decode_bytes(whole[:-20])                       # file cut short: the samples that are there
    np.testing.assert_array_equal(got[0], s16[:-10])
    got, _ = audio.decode_bytes(wav_bytes(1, 1, 16000, 16, s16.tobytes(), data_size=0))            # streamed writer, size never patched
    np.testing.assert_array_equal(got[0], s16)
    got, _ = audio.decode_bytes(wav_bytes(1, 1, 16000, 16, s16.tobytes(), data_size=0xFFFFFFFF))
    np.testing.assert_array_equal(got[0], s16)


def test_refused_containers_are_named():
    # Ogg, and the MPEG audio forms csrc/mp3.cpp does not take (Layer II, free format), are refused by name; an MPEG header with
    # nothing behind it is corrupt data (MP3 itself is decoded since round 6: tests/test_mp3.py)
    for data, word in ((b"OggS" + bytes(60), "Ogg"), (b"ID3\x03\x00\x00\x00\x00\x00\x0a" + bytes(10) + b"\xff\xfd\x90\x00" + bytes(64), "Layer II"),
                       (b"\xff\xfb\x00\x64" + bytes(64), "free-format")):
        with pytest.raises(NotImplementedError, match=word):
            audio.decode_bytes(data)
    with pytest.raises(ValueError, match="MP3"):
        audio.decode_bytes(b"\xff\xfb\x90\x64" + bytes(64))
    with pytest.raises(ValueError, match="unrecognised"):
        audio.decode_bytes(b"hello, this is not audio")
    with pytest.raises(NotImplementedError, match="format tag 2"):
        audio.decode_bytes(wav_bytes(2, 1, 16000, 4, bytes(64)))   # ADPCM


def test_load_reads_files(tmp_path):
    x = signal(2, 800, 16, 11)
    p = tmp_path / "a.flac"
    p.write_bytes(FW.encode(x, 16, 22050, block=256))
    wave, rate = audio.load(str(p))
    assert rate == 22050
    np.testing.assert_array_equal(wave, x.astype(np.int16))
    q = tmp_path / "b.bin"
    q.write_bytes(b"nothing")
    with pytest.raises(ValueError, match="b.bin"):
        audio.load(str(q))


# ----------------------------------------------------------------------------------------------------- through the front end
@pytest.mark.gpu
def test_front_end_reads_flac_and_wide_wav(tmp_path):
    """ReverbASR.compute_feats (cli/reverb.py:119-146) on the same signal in four files: 16-bit WAV and 16-bit FLAC give the
    same int16 samples, hence bit-identical features and transcripts; 24-bit WAV / FLAC arrive as left-justified int32
    (x 256 after `.to(torch.float)`), which shifts every log-mel by 2 ln 256; a 44.1 kHz float WAV goes through the device
    resampler from float samples."""
    import math
    from oracle import fbank_ref, resample_ref
    from reverb_amd import synth
    from reverb_amd.reverb import load_model
    mdir = synth.write_model_dir(str(tmp_path / "m"), "tiny")
    asr = load_model(mdir, gpu=0, dtype="f32", max_chunks=4)
    pcm = synth.synth_audio(9.0, seed=12)
    wav16, flac16 = str(tmp_path / "a.wav"), str(tmp_path / "a.flac")
    synth.write_wav(wav16, pcm)
    with open(flac16, "wb") as f:
        f.write(FW.encode(pcm.astype(np.int64)[None], 16, 16000, block=4096, predictor=("fixed", 2), partition_order=3))
    f_wav = asr.compute_feats(wav16, num_mel_bins=80).numpy()
    f_flac = asr.compute_feats(flac16, num_mel_bins=80).numpy()
    np.testing.assert_array_equal(f_wav, f_flac)
    assert asr.transcribe(flac16, mode="ctc_greedy_search", format="txt") == asr.transcribe(wav16, mode="ctc_greedy_search", format="txt")

    wide = pcm.astype(np.int64) * 256 + 77                                   # a genuine 24-bit signal
    flac24, wav24 = str(tmp_path / "b.flac"), str(tmp_path / "b.wav")
    with open(flac24, "wb") as f:
        f.write(FW.encode(np.stack([wide, -wide]), 24, 16000, block=1152, stereo="mid_side"))
    with open(wav24, "wb") as f:
        f.write(wav_bytes(1, 2, 16000, 24, b"".join(int(v).to_bytes(3, "little", signed=True)
                                                      for v in np.stack([wide, -wide]).T.reshape(-1))))
    f24 = asr.compute_feats(flac24, num_mel_bins=80).numpy()
    np.testing.assert_array_equal(f24, asr.compute_feats(wav24, num_mel_bins=80).numpy())
    want = fbank_ref.fbank((wide << 8).astype(np.float32))                    # kaldi.fbank of the float waveform, channel 0
    assert f24.shape[1:] == want.shape
    assert np.abs(f24[0] - want).max() < 5e-3
    assert abs(float(np.median(f24[0] - f_wav[0])) - 2 * math.log(65536.0)) < 0.05

    t = np.arange(int(2.0 * 44100)) / 44100.0
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t)).astype("<f4")
    wavf = str(tmp_path / "c.wav")
    with open(wavf, "wb") as f:
        f.write(wav_bytes(3, 1, 44100, 32, x.tobytes()))
    ff = asr.compute_feats(wavf, num_mel_bins=80).numpy()
    y = resample_ref.resample(x, 44100, 16000)
    got = asr.engine.waveform()
    assert got.shape == y.shape and np.abs(got - y).max() < 1e-5
    want = fbank_ref.fbank(y)
    assert ff.shape[1:] == want.shape and np.abs(ff[0] - want).max() < 5e-3


@pytest.mark.gpu
def test_front_end_transcribes_an_mp3(tmp_path):
    """`reverb.transcribe("x.mp3")` -- the reference's own usage example (README.md:61-106) -- end to end: a 44.1 kHz joint-stereo
    Layer III file (written by tests/mp3_writer.py, with an Info + LAME tag frame) is decoded on the host, channel 0 goes to the
    device as float32 (what torchaudio.load hands the reference for an mp3, whatever `normalize` says), is resampled there and
    transcribed; a float32 WAVE file holding the decoder's very samples gives bit-identical features and the same CTM."""
    import mp3_tables as MT
    import mp3_writer as Wr
    from reverb_amd import synth
    from reverb_amd.reverb import load_model
    mdir = synth.write_model_dir(str(tmp_path / "m"), "tiny")
    asr = load_model(mdir, gpu=0, dtype="f32", max_chunks=4)
    pcm = synth.synth_audio(6.0, seed=3).astype(np.float64) / 32768.0
    t = np.arange(int(6.0 * 44100)) / 44100.0
    x = np.interp(t, np.arange(len(pcm)) / 16000.0, pcm)                                   # the 16 kHz test signal at 44.1 kHz
    stereo = np.stack([x, 0.6 * x + 0.01 * np.sin(2 * np.pi * 500 * t)])
    D = np.array(MT.D) / 65536.0
    mp3 = str(tmp_path / "talk.mp3")
    with open(mp3, "wb") as f:
        f.write(Wr.Encoder(44100, 2, 192, D, mode=1, mode_ext=2, info_frame=(528, 700), plan=lambda g: Wr.GranuleCfg([0, 0, 1, 2, 3, 0][g % 6])).encode(stereo))
    wave, info = audio.load_with_info(mp3)
    assert info.container == "mp3" and info.channels == 2 and info.sample_rate == 44100 and wave.dtype == np.float32
    assert info.frames == -(-len(t) // 1152) * 1152 - 1057 - (700 - 529)
    snr = 10 * np.log10((x[3000:200000] ** 2).sum() / ((x[3000:200000] - wave[0][3000:200000]) ** 2).sum())
    assert snr > 20, snr        # aligned with the input by the tag's delay (29 dB measured: 96 kbit/s per channel, random scale factors)
    wav = str(tmp_path / "talk.wav")
    with open(wav, "wb") as f:
        f.write(wav_bytes(3, 1, 44100, 32, wave[0].astype("<f4").tobytes()))
    f_mp3 = asr.compute_feats(mp3, num_mel_bins=80).numpy()
    f_wav = asr.compute_feats(wav, num_mel_bins=80).numpy()
    np.testing.assert_array_equal(f_mp3, f_wav)
    ctm = asr.transcribe(mp3, mode="attention_rescoring", format="ctm")
    assert ctm == asr.transcribe(wav, mode="attention_rescoring", format="ctm").replace("talk.wav", "talk.mp3") or \
        [l.split()[2:] for l in ctm.split("\n")] == [l.split()[2:] for l in asr.transcribe(wav, mode="attention_rescoring", format="ctm").split("\n")]
    assert len(ctm.split("\n")) >= 1 and isinstance(asr.transcribe(mp3), str)
    asr.engine.close()


# ----------------------------------------------------------------------------------------------------- RF64, AIFF / AIFF-C
def ext80(x):
    """80-bit IEEE extended of a positive number (AIFF sample rates)"""
    import math
    m, e = math.frexp(x)                     # x = m * 2^e, 0.5 <= m < 1
    return struct.pack(">HQ", e - 1 + 16383, int(m * (1 << 64)))


def aiff_bytes(channels, rate, bits, payload, compression=None, offset=0, extra=b""):
    frames = len(payload) // (channels * ((bits + 7) // 8)) if compression not in (b"alaw", b"ulaw") else len(payload) // channels
    comm = struct.pack(">hIh", channels, frames, bits) + ext80(rate)
    if compression is not None:
        comm += compression + b"\x00\x00"                          # empty pascal string, padded
    ssnd = struct.pack(">II", offset, 0) + bytes(offset) + payload
    body = (b"AIFC" if compression is not None else b"AIFF")
    if compression is not None:
        body += b"FVER" + struct.pack(">II", 4, 0xA2805140)
    body += extra + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    return b"FORM" + struct.pack(">I", len(body)) + body


def test_aiff_and_aifc():
    rng = np.random.default_rng(3)
    n = 131
    s16 = rng.integers(-32768, 32768, size=(n, 2)).astype(np.int16)
    got, info = audio.decode_bytes(aiff_bytes(2, 44100, 16, s16.astype(">i2").tobytes()))
    assert info == audio.AudioInfo("aiff", "int16", 2, 44100, 16, n, False) and got.dtype == np.int16
    np.testing.assert_array_equal(got, s16.T)
    # AIFF-C: little-endian `sowt`, an odd-sized chunk in front, a data offset inside SSND
    got, _ = audio.decode_bytes(aiff_bytes(2, 16000, 16, s16.astype("<i2").tobytes(), compression=b"sowt", offset=6,
                                           extra=b"NAME" + struct.pack(">I", 3) + b"abc\x00"))
    np.testing.assert_array_equal(got, s16.T)
    s8 = rng.integers(-128, 128, size=(n, 1)).astype(np.int8)
    got, info = audio.decode_bytes(aiff_bytes(1, 8000, 8, s8.tobytes()))
    assert info.sample_format == "uint8"
    np.testing.assert_array_equal(got, (s8.T.astype(np.float32) + 128))
    s24 = rng.integers(-(1 << 23), 1 << 23, size=(n, 2))
    raw = b"".join(int(v).to_bytes(3, "big", signed=True) for v in s24.reshape(-1))
    got, info = audio.decode_bytes(aiff_bytes(2, 48000, 24, raw, compression=b"NONE"))
    assert info.sample_format == "int32"
    np.testing.assert_array_equal(got, (s24.T << 8).astype(np.float32))
    f32 = rng.standard_normal((n, 1)).astype(">f4")
    got, info = audio.decode_bytes(aiff_bytes(1, 22050, 32, f32.tobytes(), compression=b"fl32"))
    assert info.sample_format == "float32" and info.sample_rate == 22050
    np.testing.assert_array_equal(got, f32.T.astype(np.float32))
    f64 = rng.standard_normal((n, 1)).astype(">f8")
    got, info = audio.decode_bytes(aiff_bytes(1, 96000, 64, f64.tobytes(), compression=b"fl64"))
    assert info.sample_format == "float64"
    np.testing.assert_array_equal(got, f64.T.astype(np.float32))
    codes = np.arange(256, dtype=np.uint8)
    u, info = audio.decode_bytes(aiff_bytes(1, 8000, 16, codes.tobytes(), compression=b"ulaw"))
    w, _ = audio.decode_bytes(wav_bytes(7, 1, 8000, 8, codes.tobytes()))
    assert info.sample_format == "int16"
    np.testing.assert_array_equal(u, w)
    with pytest.raises(NotImplementedError, match="ima4"):
        audio.decode_bytes(aiff_bytes(1, 8000, 16, bytes(68), compression=b"ima4"))
    with pytest.raises(ValueError, match="COMM"):
        audio.decode_bytes(b"FORM" + struct.pack(">I", 4) + b"AIFF")
    with pytest.raises(NotImplementedError, match="IFF container"):
        audio.decode_bytes(b"FORM\x00\x00\x00\x208SVX" + bytes(32))


def test_rf64():
    s16 = np.arange(-300, 300, dtype="<i2")
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    ds64 = struct.pack("<QQQI", 0, s16.nbytes, len(s16), 0)
    body = b"WAVE" + b"ds64" + struct.pack("<I", len(ds64)) + ds64 + b"fmt " + struct.pack("<I", 16) + fmt + \
        b"data" + struct.pack("<I", 0xFFFFFFFF) + s16.tobytes() + b"junk after the declared samples"
    got, info = audio.decode_bytes(b"RF64" + struct.pack("<I", 0xFFFFFFFF) + body)
    assert info.container == "wav" and info.frames == len(s16)          # the 64-bit size of ds64, not the rest of the file
    np.testing.assert_array_equal(got[0], s16)


def test_flac_threads_on_small_and_odd_streams():
    """Explicit thread counts on streams too small or too irregular to split: the decoder falls back to the front-to-back walk."""
    for n, block, th in ((100, 16, 8), (3000, 256, 16), (70000, 1024, 255), (5000, 4096, 4)):
        x = signal(2, n, 16, n)
        got, info = audio.decode_bytes(FW.encode(x, 16, 16000, block=block, stereo="mid_side"), threads=th)
        np.testing.assert_array_equal(got, x.astype(np.int16))
        assert info.md5_checked and info.decode_threads >= 1
    sizes = [1000, 16, 2500, 1484] * 8
    x = signal(1, sum(sizes), 16, 5)
    got, info = audio.decode_bytes(FW.encode(x, 16, 16000, variable_blocks=sizes), threads=6)
    np.testing.assert_array_equal(got, x.astype(np.int16))
    assert info.decode_threads == 6                       # variable block size: runs are placed by their sample numbers


def test_mutated_files_are_rejected_or_decoded_never_worse():
    """A slice of the sanitizer corpus (scripts/fuzz): every mutated file either decodes or raises ValueError /
    NotImplementedError -- nothing else, for any thread count."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_corpus", os.path.join(os.path.dirname(__file__), "..", "scripts", "fuzz",
                                                                              "make_corpus.py"))
    mc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mc)
    rng, base = np.random.default_rng(7), mc.seeds()
    decoded = rejected = 0
    for it in range(1500):
        data = mc.mutate(base[it % len(base)], rng)
        try:
            wave, info = audio.decode_bytes(data, threads=it % 4)
            assert wave.shape == (info.channels, info.frames)
            decoded += 1
        except (ValueError, NotImplementedError, MemoryError):
            rejected += 1
    assert decoded > 100 and rejected > 100


def test_odd_sized_last_chunk_without_its_pad_byte():
    """an odd-sized chunk that ends exactly at the end of the file, pad byte missing, with no data / SSND chunk
    in front of it, moved the walker to n + 1; `n - pos` wrapped and the walk went on through the heap.  Both walkers must
    stop and name the missing chunk.  (Exact-size heap copies: under ASan these were heap-buffer-overflows.)"""
    fmt = b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
    body = b"WAVE" + fmt + b"LIST" + struct.pack("<I", 3) + b"abc"
    riff = b"RIFF" + struct.pack("<I", len(body)) + body
    assert len(riff) == 47
    with pytest.raises(ValueError, match="missing fmt or data"):
        audio.probe_bytes(riff)
    with pytest.raises(ValueError, match="missing fmt or data"):
        audio.decode_bytes(riff)
    form_body = b"AIFF" + b"ANNO" + struct.pack(">I", 5) + b"hello"
    aiff = b"FORM" + struct.pack(">I", len(form_body)) + form_body
    with pytest.raises(ValueError, match="missing COMM or SSND"):
        audio.probe_bytes(aiff)
    # the same odd chunk followed by real data still decodes (pad byte present)
    s16 = np.arange(-8, 8, dtype="<i2")
    body = b"WAVE" + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"data" + struct.pack("<I", 32) + s16.tobytes()
    got, _ = audio.decode_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    np.testing.assert_array_equal(got[0], s16)


def test_flac_header_cannot_request_an_absurd_allocation():
    """STREAMINFO's 36-bit sample count sized the output array before any frame was decoded: a 42-byte header
    could ask for terabytes.  The probe now bounds it by what the file's bytes can hold."""
    x = signal(1, 4096, 16, 3)
    data = bytearray(FW.encode(x, 16, 16000))
    # STREAMINFO starts at byte 8: min/max block (4), min/max frame (6), rate:20 ch:3 bps:5 total:36 -> the low 32 bits of
    # total are bytes 22..25, its top 4 bits the low nibble of byte 21
    data[8 + 13] |= 0x0F
    data[8 + 14:8 + 18] = b"\xff\xff\xff\xff"
    with pytest.raises(ValueError, match="more samples than the file can hold"):
        audio.probe_bytes(bytes(data))
    with pytest.raises(ValueError, match="more samples than the file can hold"):
        audio.decode_bytes(bytes(data))


def test_md5_from_two_threads_at_once():
    """the MD5 round constants were a lazily filled static table behind a plain bool."""
    import threading
    x = signal(2, 20000, 16, 11)
    data = FW.encode(x, 16, 16000)
    want = native(x, 16)
    bad = []

    def work():
        for _ in range(5):
            got, info = audio.decode_bytes(data)
            if not info.md5_checked or not np.array_equal(got, want):
                bad.append(1)
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
