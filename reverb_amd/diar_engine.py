"""Python host of the rvd_* C ABI (include/rvd.h): the two neural networks of the pyannote
diarization pipeline on one MI355X.  All compute is in librvb.so's HIP kernels; without the
library or without a GPU construction fails (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import DiarCfg, RvbError, fptr

DTYPES = {"f32": _lib.RVB_F32, "fp32": _lib.RVB_F32, "float32": _lib.RVB_F32, "bf16": _lib.RVB_BF16, "bfloat16": _lib.RVB_BF16, "fp8": _lib.RVB_FP8}


def _check(rc: int, what: str):
    if rc != 0:
        msg = _lib.load().rvd_last_error()
        raise RvbError(f"{what} failed ({rc}): {msg.decode('utf8', 'replace') if msg else ''}")


def _np32(v) -> Optional[np.ndarray]:
    if hasattr(v, "detach"):
        if not v.dtype.is_floating_point:
            return None
        v = v.detach().cpu().float().numpy()
    v = np.asarray(v)
    if v.dtype.kind != "f":
        return None
    return np.ascontiguousarray(v, dtype=np.float32)


class DiarEngine:
    """segmentation (PyanNet) + embedding (WeSpeaker ResNet34) networks resident on one GPU."""

    def __init__(self, cfg: dict, segmentation_sd: Dict, embedding_sd: Optional[Dict] = None, dtype: str = "bf16",
                 device: int = 0):
        self.lib = _lib.load()
        self.cfg = dict(cfg)
        c = DiarCfg()
        c.dtype = DTYPES[dtype]
        for k in ("sample_rate", "window_samples", "step_samples", "sinc_filters", "sinc_channels", "lstm_hidden",
                  "lstm_layers", "linear_dim", "linear_layers", "num_classes", "emb_dim"):
            setattr(c, k, int(cfg[k]))
        c.emb_channels = int(cfg["emb_channels"]) if embedding_sd is not None else 0
        self.dtype = dtype
        self._h = C.c_void_p()
        self.device_index = int(device)
        _check(self.lib.rvd_create(C.byref(c), device, C.byref(self._h)), "rvd_create")
        for prefix, sd in (("segmentation.", segmentation_sd), ("embedding.", embedding_sd or {})):
            for name, v in sd.items():
                a = _np32(v)
                if a is None:
                    continue
                shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
                _check(self.lib.rvd_load_tensor(self._h, (prefix + name).encode(), fptr(a), shape, a.ndim), f"rvd_load_tensor({name})")
        _check(self.lib.rvd_finalize(self._h), "rvd_finalize")
        self.frames = self.lib.rvd_frames_per_window(self._h)
        self.num_classes = int(cfg["num_classes"])
        self.n_windows = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.rvd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ audio
    def num_windows(self, n_samples: int) -> int:
        return int(self.lib.rvd_num_windows(self._h, int(n_samples)))

    def resample(self, pcm: np.ndarray, sample_rate: int) -> np.ndarray:
        """int16 mono PCM at `sample_rate` -> int16 at the model's rate (torchaudio.functional.resample's kernel, on the device)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1)
        n = C.c_int64(0)
        _check(self.lib.rvd_resample_pcm(self._h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.shape[0], int(sample_rate), None,
                                         C.byref(n)), "rvd_resample_pcm")
        out = np.empty(int(n.value), np.int16)
        _check(self.lib.rvd_resample_pcm(self._h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.shape[0], int(sample_rate),
                                         out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(n)), "rvd_resample_pcm")
        return out[:int(n.value)]

    def upload(self, pcm: np.ndarray) -> int:
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        _check(self.lib.rvd_upload_pcm(self._h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.shape[0]), "rvd_upload_pcm")
        self.n_windows = self.num_windows(pcm.shape[0])
        return self.n_windows

    def rerun_resident(self) -> int:
        """The recording of the last upload() is still in HBM: its front end again (what upload() does behind the copy)."""
        _check(self.lib.rvd_rerun_resident(self._h), "rvd_rerun_resident")
        return self.n_windows

    # ------------------------------------------------------------------ networks
    def segment(self, first: int = 0, n: Optional[int] = None, batch: int = 4096) -> np.ndarray:
        """log-probabilities over the powerset classes, [n, frames, classes]."""
        n = self.n_windows - first if n is None else n
        out = np.empty((n, self.frames, self.num_classes), np.float32)
        for b0 in range(0, n, batch):
            nb = min(batch, n - b0)
            _check(self.lib.rvd_segment(self._h, first + b0, nb, fptr(out[b0:b0 + nb])), "rvd_segment")
        return out

    def segment_classes(self, first: int = 0, n: Optional[int] = None, batch: int = 4096) -> np.ndarray:
        """argmax powerset class per frame, uint8 [n, frames] (the log-probabilities stay on the device)."""
        n = self.n_windows - first if n is None else n
        out = np.empty((n, self.frames), np.uint8)
        for b0 in range(0, n, batch):
            nb = min(batch, n - b0)
            _check(self.lib.rvd_segment(self._h, first + b0, nb, None), "rvd_segment")
            _check(self.lib.rvd_get_classes(self._h, out[b0:b0 + nb].ctypes.data_as(C.POINTER(C.c_uint8))), "rvd_get_classes")
        return out

    def tap(self, name: str, n: int) -> np.ndarray:
        width = {"sincnet": self.cfg["sinc_channels"], "lstm": 2 * self.cfg["lstm_hidden"]}[name]
        out = np.empty((n, self.frames, width), np.float32)
        _check(self.lib.rvd_get_tap(self._h, name.encode(), fptr(out)), "rvd_get_tap")
        return out

    def embed(self, windows: np.ndarray, masks: np.ndarray, batch: int = 1 << 20) -> np.ndarray:
        """one embedding per (window, mask) item: windows int64 [n], masks float32 [n, frames] -> [n, emb_dim]."""
        windows = np.ascontiguousarray(windows, dtype=np.int64)
        masks = np.ascontiguousarray(masks, dtype=np.float32)
        n = windows.shape[0]
        out = np.empty((n, int(self.cfg["emb_dim"])), np.float32)
        for b0 in range(0, n, batch):
            nb = min(batch, n - b0)
            _check(self.lib.rvd_embed(self._h, windows[b0:].ctypes.data_as(C.POINTER(C.c_int64)), fptr(masks[b0:b0 + nb]), nb,
                                      fptr(out[b0:b0 + nb])), "rvd_embed")
        return out

    def emb_fbank(self, window: int) -> np.ndarray:
        """80-bin hamming log-mel frames of one window as the ResNet sees them before mean subtraction."""
        out = np.empty((1200, 80), np.float32)
        n = C.c_int32()
        _check(self.lib.rvd_get_emb_fbank(self._h, int(window), fptr(out), C.byref(n)), "rvd_get_emb_fbank")
        return out[:n.value].copy()

    def set_linkage_workgroups(self, workgroups: int = 0):
        """Workgroups the clustering's merge loop may take (0 default = 16, 1, 2, 4, 8, 16); the dendrogram does not depend on it."""
        _check(self.lib.rvd_set_linkage_workgroups(self._h, int(workgroups)), "rvd_set_linkage_workgroups")

    def emb_windows_per_pass(self) -> int:
        """distinct windows one pass of the embedding trunk takes (rvd_emb_windows_per_pass)"""
        n = self.lib.rvd_emb_windows_per_pass(self._h)
        if n < 0:
            _check(n, "rvd_emb_windows_per_pass")
        return int(n)

    def emb_fp8(self):
        """dtype="fp8" engines (ResNet34 stages 3-4 on e4m3 operands): (state 0 off / not calibrated, 1 calibrating, 2 active;
        activation scales [32]; values clipped at +-448 so far)."""
        st, n, cl = C.c_int32(0), C.c_int32(32), C.c_uint32(0)
        sc = np.zeros(32, np.float32)
        _check(self.lib.rvd_get_emb_fp8(self._h, C.byref(st), fptr(sc), C.byref(n), C.byref(cl)), "rvd_get_emb_fp8")
        return int(st.value), sc, int(cl.value)

    def set_emb_fp8_scales(self, scales: np.ndarray) -> None:
        """install the 32 activation scales and activate the fp8 trunk (rvd_set_emb_fp8_scales): no calibration pass follows"""
        sc = np.ascontiguousarray(scales, np.float32).reshape(-1)
        _check(self.lib.rvd_set_emb_fp8_scales(self._h, fptr(sc), len(sc)), "rvd_set_emb_fp8_scales")

    def centroid_linkage(self, X: np.ndarray) -> np.ndarray:
        """scipy.cluster.hierarchy.linkage(X, method="centroid", metric="euclidean") on the GPU (fp64)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        n, d = X.shape
        Z = np.zeros((max(n - 1, 0), 4), np.float64)
        f64 = C.POINTER(C.c_double)
        _check(self.lib.rvd_centroid_linkage(self._h, X.ctypes.data_as(f64), n, d, Z.ctypes.data_as(f64)), "rvd_centroid_linkage")
        return Z

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on: bool):
        _check(self.lib.rvd_set_profiling(self._h, 1 if on else 0), "rvd_set_profiling")

    def reset_timings(self):
        _check(self.lib.rvd_reset_timings(self._h), "rvd_reset_timings")

    def timing(self, name: str):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        _check(self.lib.rvd_get_timing(self._h, name.encode(), C.byref(ms), C.byref(fl), C.byref(n)), "rvd_get_timing")
        return ms.value, fl.value, n.value
