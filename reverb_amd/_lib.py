"""ctypes binding of librvb.so (the C ABI declared in include/rvb.h).

The library is built in-tree by `reverb_amd.build` (hipcc, gfx950).  There is no CPU fallback:
if the shared object is missing, or no HIP device is present when an engine is created, the
error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librvb.so")

RVB_F32, RVB_BF16, RVB_FP8 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
NORM_LN, NORM_AFFINE = 0, 1


class RvbError(RuntimeError):
    pass


class AudioInfo(C.Structure):
    """rvb_audio_info (include/rvb.h)."""
    _fields_ = [("container", C.c_int32), ("sample_format", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_int32),
                ("bits_per_sample", C.c_int32), ("md5_checked", C.c_int32), ("frames", C.c_int64), ("decode_threads", C.c_int32),
                ("reserved", C.c_int32)]


class _SizedCfg(C.Structure):
    """A config struct whose first field carries its own size (checked by rvb_create / rvd_create)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        if "struct_size" not in kw:
            self.struct_size = C.sizeof(type(self))


class ModelCfg(_SizedCfg):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "dtype", "input_dim", "vocab", "d_model", "heads", "ffn_dim", "num_blocks", "cnn_kernel",
        "cnn_norm", "num_langs", "dec_heads", "dec_ffn_dim", "dec_blocks", "dec_r_blocks", "blank_id",
        "sos_id", "eos_id", "max_chunks", "chunk_frames", "cnn_causal")]


_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i16p = C.POINTER(C.c_int16)
_i64p = C.POINTER(C.c_int64)
_eng = C.c_void_p

# every exported symbol of include/rvb.h: name -> (restype, argtypes)
SIGNATURES = {
    "rvb_last_error": (C.c_char_p, []),
    "rvb_version": (C.c_char_p, []),
    "rvb_model_cfg_size": (C.c_int, []),
    "rvb_create": (C.c_int, [C.POINTER(ModelCfg), C.c_int, C.POINTER(_eng)]),
    "rvb_destroy": (None, [_eng]),
    "rvb_load_tensor": (C.c_int, [_eng, C.c_char_p, _f32p, _i64p, C.c_int]),
    "rvb_finalize": (C.c_int, [_eng, _f32p, C.c_int]),
    "rvb_num_frames": (C.c_int64, [C.c_int64]),
    "rvb_compute_feats": (C.c_int, [C.c_int, _f32p, C.c_int64, C.c_int, C.c_double, C.c_double, _f32p, _i64p]),
    "rvb_upload_pcm": (C.c_int, [_eng, _i16p, C.c_int64]),
    "rvb_upload_pcm_async": (C.c_int, [_eng, _i16p, C.c_int64]),
    "rvb_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    "rvb_host_free": (C.c_int, [C.c_void_p]),
    "rvb_set_decoding_chunk": (C.c_int, [_eng, C.c_int, C.c_int]),
    "rvb_set_fp8_policy": (C.c_int, [_eng, C.c_int, C.c_int, C.c_int]),
    "rvb_joint_decode": (C.c_int, [_eng, C.c_int, C.c_double, C.c_double, C.c_double]),
    "rvb_get_joint_result": (C.c_int, [_eng, C.c_int, _i32p, _i32p, _i32p, _f64p, _i32p, _f64p]),
    "rvb_get_joint_stats": (C.c_int, [_eng, _i64p, _i64p]),
    "rvb_upload_pcm_rate": (C.c_int, [_eng, _i16p, C.c_int64, C.c_int]),
    "rvb_get_waveform": (C.c_int, [_eng, _f32p, _i64p]),
    "rvb_upload_wave_f32": (C.c_int, [_eng, _f32p, C.c_int64, C.c_int]),
    "rvb_audio_probe": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(AudioInfo)]),
    "rvb_audio_decode_f32": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_int, C.POINTER(AudioInfo)]),
    "rvb_audio_decode_i16": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int, _i16p, C.c_int64, C.c_int, C.POINTER(AudioInfo)]),
    "rvb_fbank": (C.c_int, [_eng, _f32p, _i64p]),
    "rvb_encode": (C.c_int, [_eng, _f32p, C.c_int64, _i32p, C.c_int, C.c_int, C.c_int, C.c_float]),
    "rvb_stream_begin": (C.c_int, [_eng]),
    "rvb_stream_chunk": (C.c_int, [_eng, _f32p, C.c_int, C.c_int, _f32p, _i32p]),
    "rvb_stream_state": (C.c_int, [_eng, _i32p, _i32p]),
    "rvb_stream_finish": (C.c_int, [_eng, C.c_int, C.c_float]),
    "rvb_encoder_frames": (C.c_int, [_eng, _i32p]),
    "rvb_get_encoder_lens": (C.c_int, [_eng, _i32p]),
    "rvb_get_encoder_out": (C.c_int, [_eng, _f32p]),
    "rvb_get_ctc_logprobs": (C.c_int, [_eng, C.c_int, _f32p]),
    "rvb_get_ctc_topk": (C.c_int, [_eng, _f32p, _i32p]),
    "rvb_ctc_greedy": (C.c_int, [_eng, _i32p, _i32p, _i32p]),
    "rvb_ctc_prefix_beam": (C.c_int, [_eng, C.c_int]),
    "rvb_get_nbest_count": (C.c_int, [_eng, C.c_int, _i32p, _i32p]),
    "rvb_get_nbest": (C.c_int, [_eng, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p]),
    "rvb_prepare_rescoring": (C.c_int, [_eng, C.c_int]),
    "rvb_attention_rescore": (C.c_int, [_eng, C.c_double, C.c_double]),
    "rvb_attention_decode": (C.c_int, [_eng, C.c_int, C.c_float]),
    "rvb_get_attention_result": (C.c_int, [_eng, C.c_int, _i32p, _i32p, _f32p]),
    "rvb_get_rescored": (C.c_int, [_eng, C.c_int, _i32p, _f32p, _f64p, _f64p]),
    "rvb_get_rescored_batch": (C.c_int, [_eng, _i32p, _i32p, _i32p, _i32p, _f32p, _f64p, _f64p]),
    "rvb_fp8_recalibrate": (C.c_int, [_eng]),
    "rvb_get_fp8_scales": (C.c_int, [_eng, _f32p, _i32p]),
    "rvb_set_fp8_scales": (C.c_int, [_eng, _f32p, C.c_int32]),
    "rvb_get_fp8_saturation": (C.c_int, [_eng, C.POINTER(C.c_uint32), _i32p, C.c_int]),
    "rvb_get_fp8_subsample": (C.c_int, [_eng, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_int]),
    "rvb_get_rescore_stats": (C.c_int, [_eng, _i64p, _i64p]),
    "rvb_get_rescore_logp": (C.c_int, [_eng, C.c_int, C.c_int, C.c_int, _f32p]),
    "rvb_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rvb_comm_init": (C.c_int, [_eng, C.c_int, C.c_int, C.c_void_p]),
    "rvb_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rvb_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rvb_comm_time_allgather": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    "rvb_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "rvb_comm_barrier": (C.c_int, [C.c_void_p]),
    "rvb_comm_max_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "rvb_comm_allgather_topk": (C.c_int, [C.c_void_p, _eng, C.c_void_p, _i64p]),
    "rvb_comm_free": (C.c_int, [C.c_void_p]),
    "rvb_allgather_results": (C.c_int, [_eng, C.c_void_p, C.c_int64, C.c_void_p]),
    "rvb_comm_destroy": (C.c_int, [_eng]),
    "rvb_set_profiling": (C.c_int, [_eng, C.c_int]),
    "rvb_reset_timings": (C.c_int, [_eng]),
    "rvb_get_timing": (C.c_int, [_eng, C.c_char_p, _f64p, _f64p, _i64p]),
    "rvb_get_timing_bytes": (C.c_int, [_eng, C.c_char_p, _f64p]),
    "rvb_wer_counts": (C.c_int, [_i32p, C.c_int64, _i32p, C.c_int64, _i64p]),
}

# librvb_test.so (csrc/test_api.h): raw kernel / host-search hooks for tests/ and scripts/ -- not in the product library
TEST_SIGNATURES = {
    "rvb_test_gemm": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rvb_test_rownorm": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p, _f32p, C.c_int,
                                   C.c_int, C.c_int]),
    "rvb_test_conv_block32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]),
    "rvb_test_conv_s2sc": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]),
    "rvb_test_conv1": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rvb_test_conv_igemm_fp8": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_float, _f32p, _f32p, _f32p]),
    "rvb_test_joint_new": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]),
    "rvb_test_joint_free": (None, [C.c_void_p]),
    "rvb_test_joint_begin": (C.c_int, [C.c_void_p, C.c_int, _f32p, _i32p, C.c_int, C.c_float, C.c_float, _i32p, _i32p, _i32p, _i32p, _i32p,
                                       C.c_int]),
    "rvb_test_joint_finish": (C.c_int, [C.c_void_p, _f32p]),
    "rvb_test_joint_prefix": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p]),
    "rvb_test_joint_result": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _f64p, _i32p, _f64p]),
    "rvb_test_glu_dwconv": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, _f32p, _i32p, _f32p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _f32p, C.c_int]),
    "rvb_test_attention": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_int, C.c_int]),
    "rvb_test_attention_trie": (C.c_int, [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p,
                                          _i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int]),
    "rvb_test_logsoftmax_topk": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _f32p, _i32p, _f32p]),
    "rvb_test_lse_gather": (C.c_int, [_f32p, C.c_int, C.c_int, _i32p, _f32p]),
    "rvb_test_gemm_glu": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int]),
    "rvb_test_gemm_rowadd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rvb_test_mp3_decode": (C.c_int64, [C.c_char_p, C.c_int64, C.c_int, _f32p, C.c_int64, _i64p, _i64p, C.c_int]),
    "rvb_test_mp3_hybrid": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, _f32p]),
    "rvb_test_mp3_polyphase": (C.c_int, [_f32p, _f32p, _i32p, _f32p]),
    "rvb_test_mp3_window": (C.c_int, [_f32p]),
    "rvb_test_mp3_huffman": (C.c_int, [C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint8), _i32p]),
    "rvb_test_gemm_fp8": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                    C.c_float, _f32p, _f32p]),
    "rvb_test_rownorm_fp8": (C.c_int, [_f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, _f32p, _f32p,
                                       C.c_float, C.c_float, _f32p, _f32p]),
    "rvb_test_lse_gather_multi": (C.c_int, [_f32p, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _f32p]),
    "rvb_test_host_pool": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rvb_test_build_trie": (C.c_int, [_i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p,
                                      _i32p, _i32p, _i32p, _i32p, _i32p]),
    "rvb_test_fbank": (C.c_int, [_i16p, C.c_int64, _f32p]),
    "rvb_test_set_gemm_variant": (C.c_int, [C.c_int]),
    "rvb_test_set_gemm2_opts": (C.c_int, [C.c_int, C.c_int]),
    "rvb_test_gemm_timeline": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_longlong), C.c_int, _i32p]),
    "rvb_test_gemm_bench": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _f64p, _f64p]),
    "rvb_test_prefix_beam": (C.c_int, [_f32p, _i32p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p,
                                       _f64p]),
}



class DiarCfg(_SizedCfg):
    """rvd_model_cfg of include/rvd.h."""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "dtype", "sample_rate", "window_samples", "step_samples", "sinc_filters", "sinc_channels", "lstm_hidden",
        "lstm_layers", "linear_dim", "linear_layers", "num_classes", "emb_channels", "emb_dim")]


# every exported symbol of include/rvd.h (diarization networks; same shared library)
DIAR_SIGNATURES = {
    "rvd_last_error": (C.c_char_p, []),
    "rvd_model_cfg_size": (C.c_int, []),
    "rvd_emb_windows_per_pass": (C.c_int, [_eng]),
    "rvd_create": (C.c_int, [C.POINTER(DiarCfg), C.c_int, C.POINTER(_eng)]),
    "rvd_destroy": (None, [_eng]),
    "rvd_load_tensor": (C.c_int, [_eng, C.c_char_p, _f32p, _i64p, C.c_int]),
    "rvd_finalize": (C.c_int, [_eng]),
    "rvd_num_windows": (C.c_int64, [_eng, C.c_int64]),
    "rvd_frames_per_window": (C.c_int, [_eng]),
    "rvd_upload_pcm": (C.c_int, [_eng, _i16p, C.c_int64]),
    "rvd_rerun_resident": (C.c_int, [_eng]),
    "rvd_resample_pcm": (C.c_int, [_eng, _i16p, C.c_int64, C.c_int, _i16p, _i64p]),
    "rvd_segment": (C.c_int, [_eng, C.c_int64, C.c_int, _f32p]),
    "rvd_get_classes": (C.c_int, [_eng, C.POINTER(C.c_uint8)]),
    "rvd_get_tap": (C.c_int, [_eng, C.c_char_p, _f32p]),
    "rvd_embed": (C.c_int, [_eng, _i64p, _f32p, C.c_int, _f32p]),
    "rvd_get_emb_fbank": (C.c_int, [_eng, C.c_int64, _f32p, _i32p]),
    "rvd_centroid_linkage": (C.c_int, [_eng, _f64p, C.c_int, C.c_int, _f64p]),
    "rvd_set_linkage_workgroups": (C.c_int, [_eng, C.c_int]),
    "rvd_get_emb_fp8": (C.c_int, [_eng, _i32p, _f32p, _i32p, C.POINTER(C.c_uint32)]),
    "rvd_set_emb_fp8_scales": (C.c_int, [_eng, _f32p, C.c_int32]),
    "rvd_set_profiling": (C.c_int, [_eng, C.c_int]),
    "rvd_reset_timings": (C.c_int, [_eng]),
    "rvd_get_timing": (C.c_int, [_eng, C.c_char_p, _f64p, _f64p, _i64p]),
}

_lib = None
_test_lib = None
TEST_LIB_PATH = os.path.join(_HERE, "librvb_test.so")


def load_test():
    """dlopen librvb_test.so: the product's objects plus the rvb_test_* hooks (tests and tuning scripts only)."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_LIB_PATH):
        raise RvbError(f"{TEST_LIB_PATH} not found: run `python -m reverb_amd.build` (hipcc, gfx950)")
    lib = C.CDLL(TEST_LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DIAR_SIGNATURES.items()) + list(TEST_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _test_lib = lib
    return lib


def load():
    """dlopen librvb.so and declare every prototype; raises if the build is missing.
    RVB_LAB=1 (tests and tuning scripts that A/B kernel variants) hands out librvb_test.so instead: the same objects, built
    to read the tuning switches of csrc/common.h's lab_env() from the environment -- the product library ignores them."""
    global _lib
    if os.environ.get("RVB_LAB") == "1":
        return load_test()
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RvbError(f"{LIB_PATH} not found: run `python -m reverb_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback for the Reverb-ASR hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DIAR_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rvb_last_error()
        text = msg.decode('utf8', 'replace') if msg else ''
        if _test_lib is not None:            # the call may have been one of librvb_test.so's: its own thread-local message
            tmsg = _test_lib.rvb_last_error()
            if tmsg and tmsg != msg:
                text = (text + " | librvb_test: " if text else "") + tmsg.decode('utf8', 'replace')
        raise RvbError(f"{what or 'librvb'} failed ({rc}): {text}")


def fptr(a):
    """float32 C-contiguous numpy array -> float* (None passes NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(_f32p)


def iptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(_i32p)


def dptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(_f64p)
