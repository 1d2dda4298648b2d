"""Drop-in for the reference's Python API of the recognize_wav hot path
(`asr/wenet/cli/reverb.py`): `load_model`, `ReverbASR`, `get_available_models`, `download_model`
with the same signatures, keyword names and defaults -- compute goes to librvb's HIP kernels.

Differences a caller can observe (all supersets):
  * `ReverbASR(..., gpu=N)` / `load_model(model, gpu=N, dtype=...)`: the engine always runs on an
    MI355X (there is no CPU path); `gpu < 0` means device 0.
  * `transcribe(mode="ctc_greedy_search")` returns text/CTM (the reference raises, A1).
  * chunks are batched on the device; `batch_size` keeps its meaning for `feats_batcher` but does
    not limit the device batch (chunks are independent, reverb.py:148-180, so results are equal).
"""
from __future__ import annotations

import logging
import os
import shutil
from functools import partial
from itertools import chain
from math import ceil
from pathlib import Path
from typing import Generator, List, Tuple

import numpy as np

from .ctc_align import adjust_model_time_offset, ctc_align, hyps_to_ctm, hyps_to_txt
from .engine import Engine, SUPPORTED_MODES, joint_topk
from .search import DecodeResult
from .tokenizer import RevBpeTokenizer
from . import audio

_FRAME_DOWNSAMPLING_FACTOR = {"linear": 1, "conv2d": 4, "conv2d6": 6, "conv2d8": 8}
CACHED_MODELS_DIR = Path.home() / ".cache/reverb"
_MODELS = {"reverb_asr_v1": "https://huggingface.co/Revai/reverb-asr"}


def _torch():
    import torch
    return torch


class RvbASRModel:
    """Stands where the reference keeps `ASRModel` (`ReverbASR.model`): same `decode` seam
    (asr/wenet/transformer/asr_model.py:331-350), backed by the device engine."""

    def __init__(self, engine: Engine):
        self.engine = engine
        self.lsl_enc = self.lsl_dec = engine.cfg.num_langs > 0

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def sos_symbol(self):
        return self.engine.cfg.sos_id

    def eos_symbol(self):
        return self.engine.cfg.eos_id

    def decode(self, methods: List[str], speech, speech_lengths, beam_size: int, decoding_chunk_size: int = -1,
               num_decoding_left_chunks: int = -1, ctc_weight: float = 0.0, simulate_streaming: bool = False,
               reverse_weight: float = 0.0, context_graph=None, blank_id: int = 0, blank_penalty: float = 0.0,
               length_penalty: float = 0.0, infos=None, cat_embs=None, cv=None, cv_lengths=None):
        assert speech.shape[0] == speech_lengths.shape[0]
        assert decoding_chunk_size != 0
        if context_graph is not None:
            raise NotImplementedError("context biasing is out of scope")
        if blank_id != self.engine.cfg.blank_id:
            raise ValueError("blank_id differs from the model's ctc_blank_id")
        feats = speech.detach().cpu().numpy() if hasattr(speech, "detach") else np.asarray(speech)
        lens = speech_lengths.detach().cpu().numpy() if hasattr(speech_lengths, "detach") else np.asarray(speech_lengths)
        if cat_embs is not None:
            cat = cat_embs.detach().cpu().numpy() if hasattr(cat_embs, "detach") else np.asarray(cat_embs)
            cat = np.asarray(cat, dtype=np.float32)
            if cat.ndim == 2:
                # per-utterance language weights (encoder_layer.py:378-390, decoder_layer.py: `cat_embs[:, i]` scales layer i's output
                # of batch item b): the engine folds ONE weight vector into its language-specific layers (W = sum_i c_i W_i), so the
                # batch is decoded group by group, one group per distinct row, and the results go back in the caller's order
                if cat.shape[0] != feats.shape[0]:
                    raise ValueError(f"cat_embs has {cat.shape[0]} rows for a batch of {feats.shape[0]}")
                kw = dict(decoding_chunk_size=decoding_chunk_size, num_decoding_left_chunks=num_decoding_left_chunks, ctc_weight=ctc_weight,
                          simulate_streaming=simulate_streaming, reverse_weight=reverse_weight, blank_id=blank_id,
                          blank_penalty=blank_penalty, length_penalty=length_penalty)
                rows, groups = {}, []
                for b in range(cat.shape[0]):
                    key = cat[b].tobytes()
                    if key not in rows:
                        rows[key] = len(groups)
                        groups.append([])
                    groups[rows[key]].append(b)
                merged = {}
                for idx in groups:
                    part = self.decode(methods, feats[idx], lens[idx], beam_size, cat_embs=cat[idx[0]], **kw)
                    for m, res in part.items():
                        slot = merged.setdefault(m, [None] * cat.shape[0])
                        for b, r in zip(idx, res):
                            slot[b] = r
                return merged
            self.engine.set_cat_embs(cat)
        results = {}
        if simulate_streaming and decoding_chunk_size > 0:
            # asr_model.py:301-306: the encoder runs chunk by chunk with attention caches (encoder.py:343-402) on the whole
            # (zero padded) input, lengths are not consulted and every output frame is valid.  The reference's call leaves
            # cat_embs out, which its language-specific layers refuse (encoder_layer.py:379); here the embeddings set on
            # the engine apply, and each item of the batch is its own stream (the reference asserts batch 1).
            for b in range(feats.shape[0]):
                self.engine.forward_chunk_by_chunk(feats[b], decoding_chunk_size, num_decoding_left_chunks, return_output=False)
                self.engine.stream_finish(beam_size, blank_penalty, topk=joint_topk(methods, beam_size))
                part = self.engine.search(methods, ctc_weight, reverse_weight, length_penalty)
                for k, v in part.items():
                    results.setdefault(k, []).extend(v)
            return results
        self.engine.apply_decoding_chunk(decoding_chunk_size, num_decoding_left_chunks)
        mc = self.engine.cfg.max_chunks
        for s in range(0, feats.shape[0], mc):
            self.engine.encode(feats[s:s + mc], lens[s:s + mc], beam_size, blank_penalty, topk=joint_topk(methods, beam_size))
            part = self.engine.search(methods, ctc_weight, reverse_weight, length_penalty)
            for k, v in part.items():
                results.setdefault(k, []).extend(v)
        return results


class ReverbASR:
    def __init__(self, config, checkpoint, cmvn_path: str | None = None, tokenizer_symbols: str | None = None,
                 bpe_path: str | None = None, gpu: int = -1, overwrite_cmvn: bool = False, dtype: str = "bf16",
                 max_chunks: int = 64):
        import yaml
        torch = _torch()
        self.jit = False
        self.device = torch.device("cuda", max(gpu, 0))
        self.checkpoint = checkpoint
        with open(config, "r") as fin:
            self.configs = yaml.load(fin, Loader=yaml.FullLoader)
        self.configs["cmvn_conf"]["cmvn_file"] = self._make_path_absolute(
            self.configs["cmvn_conf"]["cmvn_file"], cmvn_path)
        tc = self.configs["tokenizer_conf"]
        tc["symbol_table_path"] = self._make_path_absolute(tc["symbol_table_path"], tokenizer_symbols)
        tc["bpe_path"] = self._make_path_absolute(tc["bpe_path"], bpe_path)
        if self.configs.get("tokenizer", "rev_bpe") not in ("rev_bpe", "char", "bpe"):
            raise NotImplementedError(f"tokenizer {self.configs.get('tokenizer')!r} is not supported")
        self.tokenizer = RevBpeTokenizer(tc["bpe_path"], tc["symbol_table_path"], tc.get("non_lang_syms_path"),
                                         split_with_space=tc.get("split_with_space", False), full_config=tc)
        self.blank_id = self._blank_id()
        self.configs["output_dim"] = len(self.tokenizer.symbol_table)

        # weights: the checkpoint as torch.load reads it (utils/checkpoint.py:29-46, strict=False)
        sd = torch.load(checkpoint, map_location="cpu", mmap=False)
        if "model0" in sd and isinstance(sd["model0"], dict):
            sd = sd["model0"]
        if overwrite_cmvn or "encoder.global_cmvn.mean" not in sd:
            mean, istd = load_cmvn(self.configs["cmvn_conf"]["cmvn_file"], self.configs["cmvn_conf"]["is_json_cmvn"])
            sd = dict(sd)
            sd["encoder.global_cmvn.mean"] = torch.from_numpy(mean).float()
            sd["encoder.global_cmvn.istd"] = torch.from_numpy(istd).float()
        self._sd, self._dtype, self._gpu, self._max_chunks = sd, dtype, max(gpu, 0), max_chunks
        self.engine = Engine(self.configs, sd, dtype=dtype, device=max(gpu, 0), max_chunks=max_chunks)
        self.model = RvbASRModel(self.engine)
        self.test_conf = self.configs["dataset_conf"]
        self.input_frame_length = self.test_conf["fbank_conf"]["frame_shift"]
        self.output_frame_length = self.input_frame_length * _FRAME_DOWNSAMPLING_FACTOR.get(
            self.configs["encoder_conf"]["input_layer"], 4)

    # ------------------------------------------------------------------ helpers
    def _blank_id(self) -> int:
        cc = self.configs.setdefault("ctc_conf", {})
        table = self.tokenizer.symbol_table
        if "<blank>" in table:
            if "ctc_blank_id" in cc:
                assert cc["ctc_blank_id"] == table["<blank>"]
            else:
                cc["ctc_blank_id"] = table["<blank>"]
        else:
            assert "ctc_blank_id" in cc, "PLZ set ctc_blank_id in yaml"
        return cc["ctc_blank_id"]

    def _make_path_absolute(self, config_path, alternate_path: str | None = None) -> str:
        if alternate_path:
            return alternate_path
        if config_path is None:
            return None
        p = Path(config_path)
        if not p.is_absolute():
            p = Path(self.checkpoint).parent / p       # assumed adjacent to the checkpoint
        return p.as_posix()

    # ------------------------------------------------------------------ front end
    def _load_pcm(self, audio_file: str, resample_rate: int):
        """-> (channel 0 in the native sample format -- int16, or float32 holding `.to(torch.float)` of anything else --, its
        sample rate); WAVE and FLAC are decoded by librvb on the host, the engine resamples to 16 kHz on the device."""
        wave, rate = audio.load(audio_file, channel=0)
        logging.info(f"detected sample rate: {rate}")
        if resample_rate != 16000:
            raise NotImplementedError("the device front end is built for a 16 kHz model (resample_rate=16000)")
        return wave[0], rate                              # kaldi.fbank uses channel 0 (Appendix A8)

    def compute_feats(self, audio_file: str, resample_rate: int = 16000, num_mel_bins=23, frame_length=25,
                      frame_shift=10, dither=0.0):
        """(1, frames, num_mel_bins) float32 tensor.  With the model's own settings (80 / 25 / 10) the same features stay resident
        in HBM for the decode that follows; any other setting (the reference passes them straight to kaldi.fbank, and its own
        default is 23 bins) goes through the stand-alone `rvb_compute_feats`."""
        if dither != 0.0:
            raise NotImplementedError("dither is random noise: the device fbank computes dither = 0.0 (what the Reverb recipe uses)")
        self.engine.upload_pcm(*self._load_pcm(audio_file, resample_rate))
        if (num_mel_bins, frame_length, frame_shift) == (80, 25, 10):
            _, feats = self.engine.fbank(return_feats=True)
            return _torch().from_numpy(feats).unsqueeze(0)
        import ctypes as C
        from ._lib import check, fptr
        wave = np.ascontiguousarray(self.engine.waveform(), np.float32)       # 16 kHz, int16 scale (resampled on the device if needed)
        n = C.c_int64(0)
        lib, dev = self.engine.lib, int(getattr(self.engine, "device_index", 0) or 0)
        check(lib.rvb_compute_feats(dev, fptr(wave), wave.size, int(num_mel_bins), float(frame_length), float(frame_shift), None, C.byref(n)),
              "rvb_compute_feats")
        feats = np.empty((n.value, int(num_mel_bins)), np.float32)
        if n.value:
            check(lib.rvb_compute_feats(dev, fptr(wave), wave.size, int(num_mel_bins), float(frame_length), float(frame_shift), fptr(feats),
                                        C.byref(n)), "rvb_compute_feats")
        return _torch().from_numpy(feats).unsqueeze(0)

    def feats_batcher(self, infeats, chunk_size: int, batch_size: int) -> Generator[Tuple, None, None]:
        """Fixed-length, non-overlapping chunks; the last one is zero padded and length-masked."""
        torch = _torch()
        nbins = self.test_conf["fbank_conf"]["num_mel_bins"]
        per_batch = chunk_size * batch_size
        total = infeats.shape[1]
        for b in range(ceil(total / per_batch)):
            piece = infeats[:, b * per_batch: (b + 1) * per_batch, :]
            nchunks = ceil(piece.shape[1] / chunk_size)
            lens = torch.full((nchunks,), chunk_size, dtype=torch.int32)
            short = nchunks * chunk_size - piece.shape[1]
            if short > 0:
                lens[-1] -= short
                piece = torch.nn.functional.pad(piece, (0, 0, 0, short, 0, 0), mode="constant", value=0)
            yield piece.reshape(-1, chunk_size, nbins), lens

    # ------------------------------------------------------------------ transcription
    def transcribe_modes(self, audio_file, modes: List[str], format: str = "txt", verbatimicity: float = 1.0,
                         chunk_size: int = 2051, batch_size: int = 1, beam_size: int = 10,
                         decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, ctc_weight: float = 0.1,
                         simulate_streaming: bool = False, reverse_weight: float = 0.0, blank_penalty: float = 0.0,
                         length_penalty: float = 0.0, timings_adjustment: float = 230) -> list[str]:
        fc = self.test_conf["fbank_conf"]
        if (fc["num_mel_bins"], fc["frame_length"], fc["frame_shift"]) != (80, 25, 10):
            raise NotImplementedError("the device fbank is built for 80 bins / 25 ms / 10 ms")
        if chunk_size < 7:
            raise ValueError("chunk_size must be at least 7 frames (Conv2dSubsampling4 needs 7 input frames, subsampling.py:201-226)")
        eng = self._engine_for_chunk(chunk_size)
        eng.upload_pcm(*self._load_pcm(audio_file, 16000))
        eng.set_cat_embs([verbatimicity, 1.0 - verbatimicity])
        if simulate_streaming and decoding_chunk_size > 0:
            # the reference's loop (cli/reverb.py:220-253) with the encoder run chunk by chunk inside model.decode
            _, feats = eng.fbank(return_feats=True)
            hyps = {m: [] for m in modes}
            for x, lens in self.feats_batcher(_torch().from_numpy(feats).unsqueeze(0), chunk_size, batch_size):
                part = self.model.decode(modes, x, lens, beam_size, decoding_chunk_size, num_decoding_left_chunks, ctc_weight,
                                         True, reverse_weight, blank_id=self.blank_id, blank_penalty=blank_penalty, length_penalty=length_penalty)
                for m in modes:
                    hyps[m].extend(part[m])
            return [get_output(format, self.tokenizer, Path(audio_file).name, hyps[mode], timings_adjustment, chunk_size,
                               self.input_frame_length, self.output_frame_length) for mode in modes]
        eng.apply_decoding_chunk(decoding_chunk_size, num_decoding_left_chunks)
        n_frames = eng.fbank()
        hyps = self.decode_resident(n_frames, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, length_penalty)
        return [get_output(format, self.tokenizer, Path(audio_file).name, hyps[mode], timings_adjustment, chunk_size,
                           self.input_frame_length, self.output_frame_length) for mode in modes]

    def _engine_for_chunk(self, chunk_size: int) -> Engine:
        """The reference accepts any --chunk_size (cli/reverb.py:188, recognize_wav.py:66-70).  The engine sizes its
        positional tables and workspace for `chunk_frames` input frames per chunk: smaller chunks run on the same
        engine, a larger one rebuilds it once (weights are re-packed from the kept state dict)."""
        if chunk_size > self.engine.cfg.chunk_frames:
            cat = self.engine._cat
            chunks = max(1, self._max_chunks * self.engine.cfg.chunk_frames // chunk_size)     # same workspace budget
            self.engine.close()
            self.engine = Engine(self.configs, self._sd, dtype=self._dtype, device=self._gpu, max_chunks=chunks,
                                 chunk_frames=chunk_size, cat_embs=cat)
            self.model = RvbASRModel(self.engine)
        return self.engine

    def decode_resident(self, n_frames: int, modes, chunk_size: int, beam_size: int, ctc_weight: float,
                        reverse_weight: float, blank_penalty: float = 0.0, length_penalty: float = 0.0):
        return self.engine.decode_resident(n_frames, modes, chunk_size, beam_size, ctc_weight, reverse_weight,
                                           blank_penalty, length_penalty)

    def transcribe(self, audio_file, mode: str = "ctc_prefix_beam_search", format: str = "txt",
                   verbatimicity: float = 1.0, chunk_size: int = 2051, batch_size: int = 1, beam_size: int = 10,
                   decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, ctc_weight: float = 0.1,
                   simulate_streaming: bool = False, reverse_weight: float = 0.0, blank_penalty: float = 0.0,
                   length_penalty: float = 0.0, timings_adjustment: float = 230) -> str:
        return self.transcribe_modes(
            audio_file, modes=[mode], format=format, verbatimicity=verbatimicity, chunk_size=chunk_size,
            batch_size=batch_size, beam_size=beam_size, decoding_chunk_size=decoding_chunk_size,
            num_decoding_left_chunks=num_decoding_left_chunks, ctc_weight=ctc_weight,
            simulate_streaming=simulate_streaming, reverse_weight=reverse_weight, blank_penalty=blank_penalty,
            length_penalty=length_penalty, timings_adjustment=timings_adjustment)[0]


def load_cmvn(cmvn_file: str, is_json: bool):
    """mean / inverse std from accumulated statistics (asr/wenet/utils/cmvn.py:21-93)."""
    import json
    import math
    if is_json:
        with open(cmvn_file) as f:
            st = json.load(f)
        sums, sqs, count = st["mean_stat"], st["var_stat"], st["frame_num"]
    else:
        with open(cmvn_file) as f:
            arr = f.read().split()
        assert arr[0] == "[" and arr[-2] == "0" and arr[-1] == "]"
        dim = int((len(arr) - 4) / 2)
        sums = [float(x) for x in arr[1:dim + 1]]
        count = float(arr[dim + 1])
        sqs = [float(x) for x in arr[dim + 2:2 * dim + 2]]
    mean, istd = [], []
    for s, q in zip(sums, sqs):
        m = s / count
        var = max(q / count - m * m, 1.0e-20)
        mean.append(m)
        istd.append(1.0 / math.sqrt(var))
    return np.array(mean), np.array(istd)


def get_output(format: str, tokenizer, audio_name: str, hyps: List[DecodeResult], timings_adjustment_ms,
               chunk_size: int, input_frame_length: int, output_frame_length: int) -> str:
    """DecodeResults of consecutive chunks -> one TXT / CTM string (cli/reverb.py:298-327)."""
    if format == "txt":
        render, sep = hyps_to_txt, " "
    elif format == "ctm":
        render, sep = partial(hyps_to_ctm, audio_name), "\n"
    else:
        raise ValueError("Invalid output format.")
    lines, shift_ms = [], 0
    for hyp in hyps:
        times = hyp.times if hyp.times is not None else hyp.ctc_frames
        if times is None:
            # `attention` mode carries no timestamps (search.py:357-360: DecodeResult(hyp.tolist())); the reference's
            # get_output then fails in ctc_align on len(None).  Text output does not need times; CTM cannot be written.
            if format != "txt":
                raise ValueError("this decoding mode produces no timestamps: use format='txt'")
            times = [0] * len(hyp.tokens)
        words = ctc_align(hyp.tokens, times, hyp.tokens_confidence, tokenizer, output_frame_length, shift_ms)
        words = adjust_model_time_offset(words, timings_adjustment_ms)
        shift_ms += chunk_size * input_frame_length
        lines.extend(list(render(words)))
    return sep.join(lines)


def load_model(model: str, gpu: int = -1, dtype: str = "bf16", max_chunks: int = 64):
    """Loads a reverb model from a directory (config.yaml + *.pt) or by name (cli/reverb.py:330-363)."""
    if Path(model).exists():
        model_dir = Path(model)
        config_path = model_dir / "config.yaml"
        checkpoint_path = list(model_dir.glob("*.pt"))[0]
    elif model in _MODELS:
        model_dir = CACHED_MODELS_DIR / model
        config_path = model_dir / "config.yaml"
        checkpoint_path = model_dir / f"{model}.pt"
        if not (model_dir.exists() and config_path.exists() and checkpoint_path.exists()):
            CACHED_MODELS_DIR.parent.mkdir(exist_ok=True, parents=True)
            shutil.rmtree(model_dir, ignore_errors=True)
            download_model(_MODELS[model], model_dir)
    else:
        raise ValueError("Please specify a local path to a model or one of our pretrained models: "
                         f"{','.join(get_available_models())}")
    config_path, checkpoint_path = config_path.resolve(), checkpoint_path.resolve()
    logging.info(f"Loading the model with {config_path = } and {checkpoint_path = }")
    return ReverbASR(str(config_path), str(checkpoint_path), gpu=gpu, dtype=dtype, max_chunks=max_chunks)


def get_available_models():
    return list(_MODELS.keys())


def download_model(url: str, root: str):
    """Clones the model repository at `url` into `root` (needs GitPython and network)."""
    from git import Repo
    Repo.clone_from(url, root)
