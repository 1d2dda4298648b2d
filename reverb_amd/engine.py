"""Python host of the librvb engine: loads a Reverb-ASR state dict into HBM through the C ABI
and exposes `ASRModel.decode`-shaped decoding (asr/wenet/transformer/asr_model.py:331-432).

PyTorch is used only to read `.pt` checkpoints; all compute is in librvb's HIP kernels."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ModelCfg, RvbError, check, dptr, fptr, iptr
from .search import DecodeResult

DTYPES = {"f32": _lib.RVB_F32, "fp32": _lib.RVB_F32, "float32": _lib.RVB_F32,
          "bf16": _lib.RVB_BF16, "bfloat16": _lib.RVB_BF16, "fp8": _lib.RVB_FP8}
SUPPORTED_MODES = ("attention", "ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring", "joint_decoding")
JOINT_PRE_BEAM_RATIO = 1.5      # transformer/search.py:458


def model_cfg_from_configs(configs: dict, dtype: str, max_chunks: int, chunk_frames: int) -> ModelCfg:
    """config.yaml -> rvb_model_cfg; refuses the model families / options the hot path does not cover
     instead of silently computing something else."""
    ec, dc = configs["encoder_conf"], configs.get("decoder_conf", {})
    if configs.get("encoder", "conformer") != "conformer":
        raise RvbError(f"encoder {configs.get('encoder')!r} is not supported (Reverb-ASR is a conformer)")
    want = dict(input_layer="conv2d", pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn",
                activation_type="swish", normalize_before=True, use_cnn_module=True, macaron_style=True)
    defaults = dict(input_layer="conv2d", pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn",
                    activation_type="swish", normalize_before=True, use_cnn_module=True, macaron_style=True)
    for k, v in want.items():
        if ec.get(k, defaults[k]) != v:
            raise RvbError(f"encoder_conf.{k}={ec.get(k)!r} is not supported (need {v!r})")
    ds = configs.get("dataset_conf", {})
    lsl = bool(ds.get("pass_cat_emb", False))
    if ds.get("add_cat_emb", False):
        raise RvbError("dataset_conf.add_cat_emb is not supported")
    nlang = int(ds["cat_emb_conf"]["emb_len"]) if lsl else 0
    vocab = int(configs["output_dim"])
    blank = int(configs.get("ctc_conf", {}).get("ctc_blank_id", 0))
    cfg = ModelCfg()
    cfg.dtype = DTYPES[dtype]
    cfg.input_dim = int(configs["input_dim"])
    cfg.vocab = vocab
    cfg.d_model = int(ec["output_size"])
    cfg.heads = int(ec["attention_heads"])
    cfg.ffn_dim = int(ec["linear_units"])
    cfg.num_blocks = int(ec["num_blocks"])
    cfg.cnn_kernel = int(ec.get("cnn_module_kernel", 15))
    cfg.cnn_norm = 0 if ec.get("cnn_module_norm", "batch_norm") == "layer_norm" else 1
    cfg.cnn_causal = 1 if ec.get("causal", False) else 0          # convolution.py:55-57: left padding cnn_kernel-1 only
    cfg.num_langs = nlang
    cfg.dec_heads = int(dc.get("attention_heads", 4))
    cfg.dec_ffn_dim = int(dc.get("linear_units", 2048))
    cfg.dec_blocks = int(dc.get("num_blocks", 0))
    cfg.dec_r_blocks = int(dc.get("r_num_blocks", 0))
    cfg.blank_id = blank
    cfg.sos_id = vocab - 1
    cfg.eos_id = vocab - 1
    cfg.max_chunks = int(max_chunks)
    cfg.chunk_frames = int(chunk_frames)
    return cfg


def joint_topk(methods, beam_size: int) -> Optional[int]:
    """CTC log-probs to keep per frame when `joint_decoding` is among the modes (its pre-beam), else None."""
    if "joint_decoding" not in methods:
        return None
    # the pre-beam itself + 8 more: log-probs that tie exactly with the pre-beam threshold are candidates too (the reference
    # compares the whole row, search.py / beam_search_timesync.py:268-270); the kernel keeps at most 64
    return min(int(JOINT_PRE_BEAM_RATIO * beam_size) + 8, 64)


def _as_numpy_f32(v) -> Optional[np.ndarray]:
    if hasattr(v, "detach"):          # torch tensor
        if not v.dtype.is_floating_point:
            return None
        v = v.detach().cpu().float().numpy()
    v = np.asarray(v)
    if v.dtype.kind != "f":
        return None
    return np.ascontiguousarray(v, dtype=np.float32)


class _PinnedBlock:
    """Owner of one rvb_host_alloc block: frees it when the last numpy view over it is garbage-collected."""

    def __init__(self, lib, ptr):
        self._free, self._ptr = lib.rvb_host_free, ptr

    def __del__(self):
        try:
            self._free(self._ptr)
        except Exception:
            pass


class Engine:
    """One librvb engine = one MI355X.  Not thread-safe (use one engine per thread/GPU)."""

    def __init__(self, configs: dict, state_dict, dtype: str = "bf16", device: int = 0, max_chunks: int = 64,
                 chunk_frames: int = 2051, cat_embs: Sequence[float] = (1.0, 0.0)):
        self.lib = _lib.load()
        self.configs = configs
        self.dtype = dtype
        self.cfg = model_cfg_from_configs(configs, dtype, max_chunks, chunk_frames)
        self.handle = C.c_void_p()
        self.device_index = int(device)
        check(self.lib.rvb_create(C.byref(self.cfg), int(device), C.byref(self.handle)), "rvb_create")
        if "model0" in state_dict and isinstance(state_dict["model0"], dict):     # checkpoint.py:38-41
            state_dict = state_dict["model0"]
        for name, value in state_dict.items():
            arr = _as_numpy_f32(value)
            if arr is None:
                continue                # e.g. BatchNorm num_batches_tracked
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(self.lib.rvb_load_tensor(self.handle, name.encode(), fptr(arr), shape, arr.ndim), f"load {name}")
        self._cat = None
        self.set_cat_embs(cat_embs)
        self.batch = 0
        self.enc_frames = 0

    # -------------------------------------------------------------------------------- lifecycle
    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.rvb_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cat_embs(self, cat_embs: Sequence[float]):
        """[verbatimicity, 1-verbatimicity] (cli/reverb.py:215-217): refolds the language-specific layers."""
        cat = np.ascontiguousarray(np.asarray(cat_embs, dtype=np.float32).reshape(-1))
        if self._cat is not None and np.array_equal(cat, self._cat):
            return
        n = self.cfg.num_langs
        if n and len(cat) != n:
            raise RvbError(f"cat_embs must have {n} entries")
        check(self.lib.rvb_finalize(self.handle, fptr(cat), len(cat) if n else 0), "rvb_finalize")
        self._cat = cat

    # -------------------------------------------------------------------------------- front end
    def pinned_pcm(self, n_samples: int) -> np.ndarray:
        """int16 array of page-locked host memory (rvb_host_alloc) for the audio reader to fill: upload_pcm from it runs
        at the PCIe rate.  The memory lives exactly as long as the array (or any view of it): it is owned by the
        array's base object and released when the last view dies -- not by close(), so an array that outlives its
        engine (ReverbASR re-creates the engine for a larger chunk_size) stays valid."""
        p = C.c_void_p()
        check(self.lib.rvb_host_alloc(C.byref(p), int(n_samples) * 2), "rvb_host_alloc")
        buf = (C.c_int16 * max(int(n_samples), 1)).from_address(p.value)
        buf._rvb_owner = _PinnedBlock(self.lib, p)        # numpy keeps `buf` alive as the base of every view
        return np.ctypeslib.as_array(buf)[:int(n_samples)]

    def upload_pcm(self, pcm: np.ndarray, sample_rate: int = 16000):
        """Mono waveform -> HBM; other rates than 16 kHz are resampled on the device (reverb.py:128-134).  int16 PCM stays
        int16; a float32 array (the `.to(torch.float)` of a source whose native format is not int16) is uploaded as it is."""
        self._n_samples_next = None             # a synchronous upload replaces a pending upload_pcm_async
        if isinstance(pcm, np.ndarray) and pcm.dtype.kind == "f":
            wave = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
            check(self.lib.rvb_upload_wave_f32(self.handle, wave.ctypes.data_as(_lib._f32p), len(wave), int(sample_rate)),
                  "rvb_upload_wave_f32")
            n = C.c_int64(0)
            check(self.lib.rvb_get_waveform(self.handle, None, C.byref(n)), "rvb_get_waveform")
            self._n_samples = int(n.value)
            return
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1)
        if sample_rate == 16000:
            check(self.lib.rvb_upload_pcm(self.handle, pcm.ctypes.data_as(_lib._i16p), len(pcm)), "rvb_upload_pcm")
            self._n_samples = len(pcm)
        else:
            check(self.lib.rvb_upload_pcm_rate(self.handle, pcm.ctypes.data_as(_lib._i16p), len(pcm), int(sample_rate)),
                  "rvb_upload_pcm_rate")
            n = C.c_int64(0)
            check(self.lib.rvb_get_waveform(self.handle, None, C.byref(n)), "rvb_get_waveform")
            self._n_samples = int(n.value)

    def upload_pcm_async(self, pcm: np.ndarray):
        """Double-buffered upload of the NEXT recording (16 kHz int16, ideally a pinned_pcm() array): returns at once, the copy
        runs underneath the decoding in progress, and the samples become the engine's audio at the next fbank().  `pcm` must
        not change until then."""
        assert isinstance(pcm, np.ndarray) and pcm.dtype == np.int16 and pcm.flags["C_CONTIGUOUS"]
        pcm = pcm.reshape(-1)
        check(self.lib.rvb_upload_pcm_async(self.handle, pcm.ctypes.data_as(_lib._i16p), len(pcm)), "rvb_upload_pcm_async")
        self._pending_pcm = pcm                 # keeps the host buffer alive until the copy has been consumed
        self._n_samples_next = len(pcm)

    def set_decoding_chunk(self, chunk_size: int = -1, num_left_chunks: int = -1):
        """Chunk mask of the encoder self-attention for the next encode() calls (<= 0: full context)."""
        check(self.lib.rvb_set_decoding_chunk(self.handle, int(chunk_size), int(num_left_chunks)), "rvb_set_decoding_chunk")

    FP8_GROUPS = {"ffn_macaron": 1, "qkv": 2, "pointwise_conv1": 4, "pointwise_conv2": 8, "ffn": 16, "subsample_conv2": 32}

    def set_fp8_policy(self, groups=None, first_block: int = 0, last_block: int = -1):
        """fp8 engines: the GEMM groups (names of FP8_GROUPS, or a bit mask) of blocks first_block..last_block that run on
        the fp8 MFMA path; None = the default (both feed-forward modules of every block)."""
        if groups is None:
            mask = -1
        elif isinstance(groups, int):
            mask = groups
        else:
            mask = sum(self.FP8_GROUPS[g] for g in set(groups))
        check(self.lib.rvb_set_fp8_policy(self.handle, int(mask), int(first_block), int(last_block)), "rvb_set_fp8_policy")

    def fp8_saturation(self, reset: bool = False) -> np.ndarray:
        """fp8 engines: values clipped at +-448 per (block, activation slot) -- [blocks, 7] in the order of fp8_scales() -- since
        the scales were calibrated / installed / last reset.  All zeros = the scales cover what was decoded."""
        n = C.c_int32(0)
        check(self.lib.rvb_get_fp8_saturation(self.handle, None, C.byref(n), 0), "rvb_get_fp8_saturation")
        out = np.zeros(max(n.value, 1), np.uint32)
        check(self.lib.rvb_get_fp8_saturation(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n), 1 if reset else 0),
              "rvb_get_fp8_saturation")
        return out[:n.value].reshape(-1, 7)

    def fp8_subsample(self, reset: bool = False):
        """fp8 engines with "subsample_conv2" in the policy: (scale of conv1's fp8 output -- 0.0 before the calibration batch --,
        values of it clipped at 448 * scale since the last reset)."""
        sc, cl = C.c_float(0.0), C.c_uint32(0)
        check(self.lib.rvb_get_fp8_subsample(self.handle, C.byref(sc), C.byref(cl), 1 if reset else 0), "rvb_get_fp8_subsample")
        return float(sc.value), int(cl.value)

    def fp8_scale_vector(self) -> Optional[np.ndarray]:
        """fp8 engines: EVERY calibrated activation scale as rvb_get_fp8_scales returns them -- 7 per conformer block, then the
        scale of conv1's fp8 output (policy group "subsample_conv2"; 0.0 = not measured) -- or None before the calibration
        batch.  This is the vector ranks of a sharded run exchange (dist.share_fp8_scales) and set_fp8_scales() installs."""
        n = C.c_int32(0)
        check(self.lib.rvb_get_fp8_scales(self.handle, None, C.byref(n)), "rvb_get_fp8_scales")
        if n.value == 0:
            return None
        out = np.empty(n.value, np.float32)
        check(self.lib.rvb_get_fp8_scales(self.handle, fptr(out), C.byref(n)), "rvb_get_fp8_scales")
        return out

    def fp8_scales(self) -> Optional[np.ndarray]:
        """fp8 engines: the calibrated activation scales of the conformer blocks [blocks, 7], or None before the calibration batch."""
        v = self.fp8_scale_vector()
        return None if v is None else v[:(v.size // 7) * 7].reshape(-1, 7)

    def set_fp8_scales(self, scales):
        """Install activation scales (e.g. the element-wise maximum over the ranks of a sharded run, or scales measured
        once): the engine counts as calibrated, the next encode runs in fp8.  Accepts [blocks, 7] (the subsampling scale is
        left as it is) or the whole fp8_scale_vector()."""
        a = np.ascontiguousarray(scales, dtype=np.float32).reshape(-1)
        check(self.lib.rvb_set_fp8_scales(self.handle, fptr(a), len(a)), "rvb_set_fp8_scales")

    def apply_decoding_chunk(self, decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1):
        """What BaseEncoder.forward does with these two arguments (encoder.py:140-145 -> add_optional_chunk_mask,
        utils/mask.py:126-197): they select a chunk mask only for models configured with use_dynamic_chunk, a
        static_chunk_size model is always chunk-masked, any other model ignores them."""
        ec = self.configs.get("encoder_conf", {})
        if ec.get("use_dynamic_chunk", False):
            if decoding_chunk_size > 0:
                self.set_decoding_chunk(decoding_chunk_size, num_decoding_left_chunks)
            else:
                self.set_decoding_chunk(-1, -1)
        elif int(ec.get("static_chunk_size", 0)) > 0:
            self.set_decoding_chunk(int(ec["static_chunk_size"]), num_decoding_left_chunks)
        else:
            self.set_decoding_chunk(-1, -1)

    def waveform(self) -> np.ndarray:
        """The waveform the fbank reads (float32, int16 scale; after resampling if any)."""
        out = np.empty(max(self._n_samples, 1), np.float32)
        n = C.c_int64(0)
        check(self.lib.rvb_get_waveform(self.handle, fptr(out), C.byref(n)), "rvb_get_waveform")
        return out[:int(n.value)]

    def fbank(self, return_feats: bool = False):
        """Kaldi fbank of the uploaded PCM; features stay resident in HBM.  Returns n_frames
        (and the (n_frames, 80) float32 array when asked)."""
        n = C.c_int64(0)
        out = None
        if getattr(self, "_n_samples_next", None) is not None:      # upload_pcm_async: this call makes those samples the audio
            self._n_samples, self._n_samples_next = self._n_samples_next, None
        if return_feats:
            out = np.empty((int(self.lib.rvb_num_frames(self._n_samples)), 80), np.float32)
        check(self.lib.rvb_fbank(self.handle, fptr(out), C.byref(n)), "rvb_fbank")
        self.n_frames = int(n.value)
        return (self.n_frames, out) if return_feats else self.n_frames

    # -------------------------------------------------------------------------------- decode
    def encode(self, feats: Optional[np.ndarray], lens, beam: int, blank_penalty: float = 0.0, first_chunk: int = 0,
               T0: Optional[int] = None, topk: Optional[int] = None):
        """`beam`: the search beam of the modes that follow; `topk` (>= beam): CTC log-probs kept per frame, when a mode needs
        more than the beam (joint_decoding's pre-beam = int(1.5 * beam))."""
        lens = np.ascontiguousarray(np.asarray(lens, dtype=np.int32).reshape(-1))
        B = len(lens)
        if feats is not None:
            feats = np.ascontiguousarray(feats, dtype=np.float32)
            assert feats.ndim == 3 and feats.shape[0] == B and feats.shape[2] == self.cfg.input_dim
            T0 = feats.shape[1]
        elif T0 is None:
            T0 = self.cfg.chunk_frames
        k = max(int(beam), int(topk or 0))
        if k > 64:
            raise RvbError(f"the CTC kernel keeps at most 64 log-probs per frame; beam {beam}" +
                           (f" with joint_decoding's pre-beam {topk}" if topk else "") + " needs more")
        check(self.lib.rvb_encode(self.handle, fptr(feats), int(first_chunk), iptr(lens), B, int(T0), k,
                                  float(blank_penalty)), "rvb_encode")
        t = C.c_int32(0)
        check(self.lib.rvb_encoder_frames(self.handle, C.byref(t)))
        self.batch, self.enc_frames, self.beam, self.topk = B, int(t.value), int(beam), k
        # joint_decode's tie retry encodes the batch again; only an encode that asked for joint decoding's pre-beam (topk) and
        # so may be followed by joint_decode keeps its arguments, and joint_decode drops them when it is done (every encode pinned
        # hundreds of MB of features for the engine's lifetime)
        self._last_encode = ((feats, lens, int(beam), float(blank_penalty), int(first_chunk), int(T0))
                             if topk and k < 64 else None)      # feats None: the resident features (valid until the next upload)

    # -------------------------------------------------------------------------------- streaming encoder
    def stream_begin(self):
        """Empty the attention cache: the next forward_chunk starts a new stream (encoder.py:385-388)."""
        check(self.lib.rvb_stream_begin(self.handle), "rvb_stream_begin")

    def forward_chunk(self, xs: np.ndarray, required_cache_size: int = -1, return_output: bool = True):
        """BaseEncoder.forward_chunk (encoder.py:231-341) on the engine's stream state: xs (time, 80) raw log-mel of one
        chunk -> (chunk_frames_out, d) encoder output; `offset` and the attention cache are kept by the engine
        (stream_state())."""
        xs = np.ascontiguousarray(xs, dtype=np.float32).reshape(-1, self.cfg.input_dim)
        cap = max((xs.shape[0] - 3) // 2 + 1, 1)
        out = np.empty((cap, self.cfg.d_model), np.float32) if return_output else None
        n = C.c_int32(0)
        check(self.lib.rvb_stream_chunk(self.handle, fptr(xs), xs.shape[0], int(required_cache_size), fptr(out), C.byref(n)),
              "rvb_stream_chunk")
        return out[:n.value] if return_output else n.value

    def stream_state(self):
        """(offset, cached frames): encoder frames produced so far / frames in the attention cache (att_cache.size(2))."""
        o, c = C.c_int32(0), C.c_int32(0)
        check(self.lib.rvb_stream_state(self.handle, C.byref(o), C.byref(c)))
        return int(o.value), int(c.value)

    def forward_chunk_by_chunk(self, xs: np.ndarray, decoding_chunk_size: int, num_decoding_left_chunks: int = -1,
                               return_output: bool = True):
        """BaseEncoder.forward_chunk_by_chunk (encoder.py:343-402): overlapping input windows of (chunk-1)*4 + 7 frames every
        4*chunk frames, attention cache of chunk * left frames.  xs (T, 80) -> (T', d)."""
        assert decoding_chunk_size > 0
        xs = np.ascontiguousarray(xs, dtype=np.float32).reshape(-1, self.cfg.input_dim)
        subsampling, context = 4, 7                       # Conv2dSubsampling4: rate 4, right_context 6 (+ current frame)
        stride = subsampling * decoding_chunk_size
        window = (decoding_chunk_size - 1) * subsampling + context
        required = decoding_chunk_size * num_decoding_left_chunks
        self.stream_begin()
        outs = []
        for cur in range(0, xs.shape[0] - context + 1, stride):
            y = self.forward_chunk(xs[cur:min(cur + window, xs.shape[0])], required, return_output)
            if return_output:
                outs.append(y)
        if not return_output:
            return None
        return np.concatenate(outs) if outs else np.zeros((0, self.cfg.d_model), np.float32)

    def stream_finish(self, beam: int, blank_penalty: float = 0.0, topk: Optional[int] = None):
        """CTC head + top-k over the streamed frames; greedy() / prefix_beam() / rescore() / encoder_out() then see the
        stream as one chunk."""
        k = max(int(beam), int(topk or 0))
        check(self.lib.rvb_stream_finish(self.handle, k, float(blank_penalty)), "rvb_stream_finish")
        t = C.c_int32(0)
        check(self.lib.rvb_encoder_frames(self.handle, C.byref(t)))
        self.batch, self.enc_frames, self.beam, self.topk = 1, int(t.value), int(beam), k

    def encoder_lens(self) -> np.ndarray:
        out = np.empty(self.batch, np.int32)
        check(self.lib.rvb_get_encoder_lens(self.handle, iptr(out)))
        return out

    def encoder_out(self) -> np.ndarray:
        out = np.empty((self.batch, self.enc_frames, self.cfg.d_model), np.float32)
        check(self.lib.rvb_get_encoder_out(self.handle, fptr(out)))
        return out

    def ctc_logprobs(self, chunk: int) -> np.ndarray:
        out = np.empty((self.enc_frames, self.cfg.vocab), np.float32)
        check(self.lib.rvb_get_ctc_logprobs(self.handle, int(chunk), fptr(out)))
        return out

    def ctc_topk(self):
        k = getattr(self, "topk", self.beam)
        v = np.empty((self.batch, self.enc_frames, k), np.float32)
        i = np.empty((self.batch, self.enc_frames, k), np.int32)
        check(self.lib.rvb_get_ctc_topk(self.handle, fptr(v), iptr(i)))
        return v, i

    def greedy(self) -> List[DecodeResult]:
        T = self.enc_frames
        tok = np.empty((self.batch, T), np.int32); n = np.empty(self.batch, np.int32); fr = np.empty((self.batch, T), np.int32)
        check(self.lib.rvb_ctc_greedy(self.handle, iptr(tok), iptr(n), iptr(fr)), "rvb_ctc_greedy")
        res = []
        for b in range(self.batch):
            r = DecodeResult(tok[b, :n[b]].tolist())
            r.ctc_frames = fr[b, :n[b]].tolist()
            res.append(r)
        return res

    def _nbest(self, chunk: int):
        nh, ml = C.c_int32(0), C.c_int32(0)
        check(self.lib.rvb_get_nbest_count(self.handle, chunk, C.byref(nh), C.byref(ml)))
        nh, ml = nh.value, max(ml.value, 1)
        tok = np.empty((nh, ml), np.int32); lens = np.empty(nh, np.int32)
        tim = np.empty((nh, ml), np.int32); tl = np.empty(nh, np.int32); sc = np.empty(nh, np.float64)
        check(self.lib.rvb_get_nbest(self.handle, chunk, iptr(tok), iptr(lens), iptr(tim), iptr(tl), dptr(sc)))
        nbest = [tuple(tok[i, :lens[i]].tolist()) for i in range(nh)]
        times = [tim[i, :tl[i]].tolist() for i in range(nh)]
        return nbest, sc.tolist(), times

    def prefix_beam(self) -> List[DecodeResult]:
        check(self.lib.rvb_ctc_prefix_beam(self.handle, self.beam), "rvb_ctc_prefix_beam")
        res = []
        for b in range(self.batch):
            nbest, scores, times = self._nbest(b)
            res.append(DecodeResult(tokens=nbest[0], score=scores[0], times=times[0], nbest=nbest, nbest_scores=scores,
                                    nbest_times=times))
        return res

    def rescore(self, prefix_results: List[DecodeResult], ctc_weight: float, reverse_weight: float) -> List[DecodeResult]:
        check(self.lib.rvb_attention_rescore(self.handle, float(ctc_weight), float(reverse_weight)), "rvb_attention_rescore")
        res = []
        for b, pr in enumerate(prefix_results):
            bi, sc, cf = C.c_int32(0), C.c_float(0), C.c_double(0)
            check(self.lib.rvb_get_rescored(self.handle, b, C.byref(bi), C.byref(sc), C.byref(cf), None))
            tc = np.empty(max(len(pr.nbest[bi.value]), 1), np.float64)
            check(self.lib.rvb_get_rescored(self.handle, b, None, None, None, dptr(tc)))
            res.append(DecodeResult(pr.nbest[bi.value], float(sc.value), confidence=float(cf.value),
                                    times=pr.nbest_times[bi.value],
                                    tokens_confidence=tc[:len(pr.nbest[bi.value])].tolist()))
        return res

    def _rescore_bulk(self, ctc_weight: float, reverse_weight: float) -> List[DecodeResult]:
        check(self.lib.rvb_attention_rescore(self.handle, float(ctc_weight), float(reverse_weight)), "rvb_attention_rescore")
        return self._rescore_fetch()

    def _rescore_fetch(self) -> List[DecodeResult]:
        """The winners of the last rvb_attention_rescore in one bulk read."""
        B, T = self.batch, self.enc_frames
        lens = np.empty(B, np.int32); tok = np.empty((B, T), np.int32); tl = np.empty(B, np.int32)
        tim = np.empty((B, T), np.int32); sc = np.empty(B, np.float32); cf = np.empty(B, np.float64)
        tc = np.empty((B, T), np.float64)
        check(self.lib.rvb_get_rescored_batch(self.handle, iptr(lens), iptr(tok), iptr(tl), iptr(tim), fptr(sc), dptr(cf),
                                              dptr(tc)), "rvb_get_rescored_batch")
        return [DecodeResult(tuple(tok[b, :lens[b]].tolist()), float(sc[b]), confidence=float(cf[b]),
                             times=tim[b, :tl[b]].tolist(), tokens_confidence=tc[b, :lens[b]].tolist()) for b in range(B)]

    def rescore_stats(self):
        """(decoder rows computed, (hypothesis, position) pairs served) of the last rescoring: distinct prefixes vs the
        padded batch the reference runs."""
        r, p = C.c_int64(0), C.c_int64(0)
        check(self.lib.rvb_get_rescore_stats(self.handle, C.byref(r), C.byref(p)))
        return int(r.value), int(p.value)

    def rescore_logp(self, chunk: int, hyp: int, length: int, right: bool = False) -> np.ndarray:
        out = np.empty(length + 1, np.float32)
        check(self.lib.rvb_get_rescore_logp(self.handle, chunk, hyp, 1 if right else 0, fptr(out)))
        return out

    def attention_beam(self, length_penalty: float = 0.0) -> List[DecodeResult]:
        """`attention` mode (search.py:251-360): tokens only, like the reference's DecodeResult(hyp.tolist())."""
        check(self.lib.rvb_attention_decode(self.handle, self.beam, float(length_penalty)), "rvb_attention_decode")
        tok = np.empty(max(self.enc_frames, 1), np.int32)
        res = []
        for b in range(self.batch):
            n, sc = C.c_int32(0), C.c_float(0)
            check(self.lib.rvb_get_attention_result(self.handle, b, iptr(tok), C.byref(n), C.byref(sc)), "rvb_get_attention_result")
            res.append(DecodeResult(tok[:n.value].tolist()))
        return res

    def joint_decode(self, ctc_weight: float, length_bonus: float = 0.0, pre_beam_ratio: float = JOINT_PRE_BEAM_RATIO
                     ) -> List[DecodeResult]:
        """`joint_decoding` (search.py:450-496): time-synchronous joint CTC / attention beam search of the last encoded batch
        (encode(..., topk=int(pre_beam_ratio * beam)) first).  DecodeResult: tokens, joint score, start frame and confidence
        per token -- the fields the reference fills -- plus `end_times`."""
        rc = self.lib.rvb_joint_decode(self.handle, self.beam, float(ctc_weight), float(pre_beam_ratio), float(length_bonus))
        if (rc == -5 and self.topk < 64 and getattr(self, "_last_encode", None) is not None
                and b"tie exactly" in self.lib.rvb_last_error()):
            # RVB_E_UNSUPPORTED here = a frame holds a longer run of log-probs that tie EXACTLY with the pre-beam threshold than
            # the kept top-k covers (the reference compares the whole row, beam_search_timesync.py:268-270).  Rather than fail
            # the decode after all the work is done, encode the batch again keeping the kernel's maximum of 64
            # log-probs per frame and search once more; only a run longer than that is refused.
            feats, lens, beam, blank_penalty, first_chunk, T0 = self._last_encode
            self.encode(feats, lens, beam, blank_penalty, first_chunk, T0, topk=64)
            rc = self.lib.rvb_joint_decode(self.handle, self.beam, float(ctc_weight), float(pre_beam_ratio), float(length_bonus))
        self._last_encode = None
        check(rc, "rvb_joint_decode")
        T = max(self.enc_frames, 1)
        tok = np.empty(T, np.int32); st = np.empty(T, np.int32); en = np.empty(T, np.int32); cf = np.empty(T, np.float64)
        res = []
        for b in range(self.batch):
            n, sc = C.c_int32(0), C.c_double(0)
            check(self.lib.rvb_get_joint_result(self.handle, b, iptr(tok), iptr(st), iptr(en), dptr(cf), C.byref(n), C.byref(sc)),
                  "rvb_get_joint_result")
            k = n.value
            r = DecodeResult(tok[:k].tolist(), float(sc.value), times=st[:k].tolist(), tokens_confidence=cf[:k].tolist())
            r.end_times = en[:k].tolist()
            res.append(r)
        return res

    def joint_stats(self):
        """(decoder rows computed, batched decoder steps) of the last joint_decode."""
        r, p = C.c_int64(0), C.c_int64(0)
        check(self.lib.rvb_get_joint_stats(self.handle, C.byref(r), C.byref(p)))
        return int(r.value), int(p.value)

    def search(self, methods: Sequence[str], ctc_weight: float, reverse_weight: float, length_penalty: float = 0.0
               ) -> Dict[str, List[DecodeResult]]:
        """Search stages of ASRModel.decode (asr_model.py:391-432) on the last encoded batch."""
        results: Dict[str, List[DecodeResult]] = {}
        for m in methods:
            if m not in SUPPORTED_MODES:
                raise RvbError(f"decoding mode {m!r} is not built yet (supported: {', '.join(SUPPORTED_MODES)})")
        if "attention" in methods:
            results["attention"] = self.attention_beam(length_penalty)
        if "joint_decoding" in methods:          # asr_model.py:427-431: length_bonus = length_penalty
            results["joint_decoding"] = self.joint_decode(ctc_weight, length_penalty)
        if "ctc_greedy_search" in methods:
            results["ctc_greedy_search"] = self.greedy()
        if "attention_rescoring" in methods and "ctc_prefix_beam_search" not in methods:
            # fast path: n-best lists stay in the engine, one bulk read of the winners; the decoders' memory keys / values are
            # enqueued first so that the device has work while the host searches the last slice
            check(self.lib.rvb_prepare_rescoring(self.handle, 1 if reverse_weight > 0 else 0), "rvb_prepare_rescoring")
            check(self.lib.rvb_ctc_prefix_beam(self.handle, self.beam), "rvb_ctc_prefix_beam")
            results["attention_rescoring"] = self._rescore_bulk(ctc_weight, reverse_weight)
        elif "ctc_prefix_beam_search" in methods or "attention_rescoring" in methods:
            pref = self.prefix_beam()
            if "ctc_prefix_beam_search" in methods:
                results["ctc_prefix_beam_search"] = pref
            if "attention_rescoring" in methods:
                results["attention_rescoring"] = self.rescore(pref, ctc_weight, reverse_weight)
        return results

    def decode_resident(self, n_frames: int, modes, chunk_size: int, beam_size: int, ctc_weight: float,
                        reverse_weight: float, blank_penalty: float = 0.0, length_penalty: float = 0.0
                        ) -> Dict[str, List[DecodeResult]]:
        """Decode the device-resident features of the last fbank() call: fixed, non-overlapping
        chunks with a length-masked zero-padded tail (feats_batcher, cli/reverb.py:148-180),
        `max_chunks` chunks per launch, results concatenated in chunk order."""
        if not 7 <= chunk_size <= self.cfg.chunk_frames:
            raise RvbError(f"resident decoding needs 7 <= chunk_size <= engine chunk_frames ({self.cfg.chunk_frames}); "
                           "ReverbASR rebuilds the engine for larger chunks")
        n_chunks = -(-n_frames // chunk_size)
        lens = np.full(n_chunks, chunk_size, np.int32)
        if n_chunks:
            lens[-1] = n_frames - (n_chunks - 1) * chunk_size
        out: Dict[str, List[DecodeResult]] = {m: [] for m in modes}
        for s in range(0, n_chunks, self.cfg.max_chunks):
            e = min(n_chunks, s + self.cfg.max_chunks)
            self.encode(None, lens[s:e], beam_size, blank_penalty, first_chunk=s, T0=chunk_size, topk=joint_topk(modes, beam_size))
            part = self.search(modes, ctc_weight, reverse_weight, length_penalty)
            for m in modes:
                out[m].extend(part[m])
        return out

    # -------------------------------------------------------------------------------- timings
    def set_profiling(self, on, gemm_only: bool = False):
        """HIP-event stage timing: every stage, or (gemm_only) just the GEMM launches the roofline figure needs."""
        check(self.lib.rvb_set_profiling(self.handle, 0 if not on else 2 if gemm_only else 1))

    def reset_timings(self):
        check(self.lib.rvb_reset_timings(self.handle))

    def timing(self, name: str):
        ms, fl, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        check(self.lib.rvb_get_timing(self.handle, name.encode(), C.byref(ms), C.byref(fl), C.byref(n)))
        by = C.c_double(0)
        check(self.lib.rvb_get_timing_bytes(self.handle, name.encode(), C.byref(by)))
        return {"ms": ms.value, "flops": fl.value, "launches": n.value, "bytes": by.value}
