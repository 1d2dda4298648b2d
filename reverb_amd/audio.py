"""Audio file reader of the front end: `torchaudio.load(audio_file, normalize=False)` of the reference
(asr/wenet/cli/reverb.py:128) for RIFF/WAVE, AIFF, FLAC and MP3 (MPEG audio Layer III, csrc/mp3.cpp: float32 in [-1, 1], as
torchaudio returns it whatever `normalize` says), decoded by librvb on the host (csrc/audio.cpp, include/rvb.h
`rvb_audio_*`).  normalize=False keeps the decoder's native sample format; the array returned here has the dtype
and the values of the tensor torchaudio hands back (int16 for 16-bit PCM / A-law / mu-law / FLAC up to 16 bits,
float32 holding the `.to(torch.float)` values for everything else -- uint8, left-justified int32, float)."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Tuple

import numpy as np

from . import _lib

CONTAINERS = {1: "wav", 2: "flac", 3: "aiff", 4: "mp3"}
SAMPLE_FORMATS = {1: "uint8", 2: "int16", 3: "int32", 4: "float32", 5: "float64"}


class AudioInfo(NamedTuple):
    container: str
    sample_format: str        # dtype of the tensor torchaudio.load(normalize=False) returns
    channels: int
    sample_rate: int
    bits_per_sample: int
    frames: int
    md5_checked: bool         # FLAC: the STREAMINFO signature was present and matched the decoded PCM
    decode_threads: int = 1


def _check(rc, lib, what):
    if rc < 0:
        msg = lib.rvb_last_error().decode() or what
        raise (NotImplementedError if rc == -5 else ValueError)(msg)
    return rc


def _info(raw) -> AudioInfo:
    return AudioInfo(CONTAINERS[raw.container], SAMPLE_FORMATS[raw.sample_format], raw.channels, raw.sample_rate,
                     raw.bits_per_sample, int(raw.frames), bool(raw.md5_checked), max(int(raw.decode_threads), 1))


def probe_bytes(data: bytes) -> AudioInfo:
    lib = _lib.load()
    raw = _lib.AudioInfo()
    _check(lib.rvb_audio_probe(data, len(data), C.byref(raw)), lib, "rvb_audio_probe")
    return _info(raw)


def decode_bytes(data: bytes, channel: int = -1, verify_md5: bool = True, threads: int = 0) -> Tuple[np.ndarray, AudioInfo]:
    """-> (array of shape (channels, frames) -- (1, frames) when one `channel` is asked for --, AudioInfo).  int16 when the
    native format is int16, else float32 with the values `.to(torch.float)` gives.  FLAC: `verify_md5` checks the decoded
    PCM against the file's signature; `threads` host threads decode runs of frames (0 = by size, at most 16)."""
    flags = (0 if verify_md5 else 1) | (int(threads) & 0xFF) << 8
    lib = _lib.load()
    raw = _lib.AudioInfo()
    _check(lib.rvb_audio_probe(data, len(data), C.byref(raw)), lib, "rvb_audio_probe")
    rows = raw.channels if channel < 0 else 1
    if raw.sample_format == 2:
        out = np.empty((rows, raw.frames), np.int16)
        rc = lib.rvb_audio_decode_i16(data, len(data), channel, out.ctypes.data_as(_lib._i16p), out.size, flags, C.byref(raw))
    else:
        out = np.empty((rows, raw.frames), np.float32)
        rc = lib.rvb_audio_decode_f32(data, len(data), channel, out.ctypes.data_as(_lib._f32p), out.size, flags, C.byref(raw))
    _check(rc, lib, "rvb_audio_decode")
    return out, _info(raw)


def load_with_info(path: str, channel: int = -1) -> Tuple[np.ndarray, AudioInfo]:
    with open(path, "rb") as f:
        data = f.read()
    try:
        return decode_bytes(data, channel)
    except (ValueError, NotImplementedError) as e:
        raise type(e)(f"{path}: {e}") from None


def load(path: str, channel: int = -1) -> Tuple[np.ndarray, int]:
    """`torchaudio.load(path, normalize=False)` -> (waveform (channels, frames), sample_rate)."""
    wave, info = load_with_info(path, channel)
    return wave, info.sample_rate


def normalized(wave: np.ndarray, info: AudioInfo) -> np.ndarray:
    """The float32 waveform in [-1, 1) that `torchaudio.load(path)` (normalize=True, what pyannote's Audio reads) returns."""
    w = wave.astype(np.float32)
    if info.sample_format == "int16":
        return w / np.float32(32768.0)
    if info.sample_format == "int32":
        return w / np.float32(2147483648.0)
    if info.sample_format == "uint8":
        return (w - np.float32(128.0)) / np.float32(128.0)
    return w
