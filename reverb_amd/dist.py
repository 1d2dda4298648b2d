"""Chunk-sharded multi-GPU decoding: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), no data-path collective, one all-gather of the per-chunk results at the end.

The reference has no multi-GPU inference; what makes the shard exact is that its
long-form driver carries no state between chunks: `feats_batcher` cuts fixed, non-overlapping
chunks and each batch is decoded on its own (asr/wenet/cli/reverb.py:148-180, 220-253); only the
per-chunk time offset (`:320-325`) couples them, on the host.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import List, Optional, Tuple

import numpy as np

from .search import DecodeResult

FRAME_SHIFT, FRAME_LEN = 160, 400      # samples (10 ms / 25 ms at 16 kHz)


def chunk_ranges(n_chunks: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous chunk range [c0, c1) per rank; the first `n_chunks % world` ranks get one more."""
    base, extra = divmod(n_chunks, world)
    out, c = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append((c, c + k))
        c += k
    return out


def num_frames(n_samples: int) -> int:
    return 0 if n_samples < FRAME_LEN else 1 + (n_samples - FRAME_LEN) // FRAME_SHIFT


def sample_range(n_samples: int, chunk_frames: int, c0: int, c1: int) -> Tuple[int, int]:
    """Samples a rank needs for chunks [c0, c1): frame i covers samples [160 i, 160 i + 400), so the
    slice starts at the first frame of chunk c0 and carries a 240-sample right halo."""
    total = num_frames(n_samples)
    f0, f1 = min(c0 * chunk_frames, total), min(c1 * chunk_frames, total)
    if f1 <= f0:
        return 0, 0
    return f0 * FRAME_SHIFT, (f1 - 1) * FRAME_SHIFT + FRAME_LEN


# ---- result exchange ------------------------------------------------------------------------------------------
# One rank's results travel as ONE flat int32 buffer (ragged rows, no padding to a longest row):
#   [MAGIC, count, n_int_words, n_doubles, total_tokens, 0, 0, 0]                      header (8 words)
#   per row: k, kt, kf, kc, tokens[k], times[kt], ctc_frames[kf]                         (-1 = the field is None)
#   (pad to an even word), then the doubles as int32 pairs: per row score, confidence, tokens_confidence[kc]
# so that a single all-gather of a fixed-capacity buffer carries everything.  The capacity is
# `max_count * _ROW_WORDS` words; a rank whose rows do not fit says so in its header (n_int_words / n_doubles are
# the sizes it NEEDS), every rank sees that after the gather, raises the (process-sticky) row capacity by the same
# rule and repeats -- the common case is one collective, an overflow costs one more, once.
_MAGIC, _HDR = 0x52564231, 8
_ROW_WORDS = 512                 # int32 words reserved per result row; grows on overflow (same on every rank)


def pack_results(hyps: Sequence[DecodeResult]) -> np.ndarray:
    ints: List[int] = []
    dbl: List[float] = []
    total = 0
    for h in hyps:
        tok = list(h.tokens)
        tim, frm, tc = h.times, getattr(h, "ctc_frames", None), h.tokens_confidence
        ints += (len(tok), -1 if tim is None else len(tim), -1 if frm is None else len(frm), -1 if tc is None else len(tc))
        ints += tok
        if tim:
            ints += tim
        if frm:
            ints += frm
        dbl += (h.score, h.confidence)
        if tc:
            dbl += tc
        total += len(tok)
    if len(ints) & 1:
        ints.append(0)
    out = np.empty(_HDR + len(ints) + 2 * len(dbl), np.int32)
    out[:_HDR] = (_MAGIC, len(hyps), len(ints), len(dbl), total, 0, 0, 0)
    out[_HDR:_HDR + len(ints)] = ints
    out[_HDR + len(ints):] = np.asarray(dbl, np.float64).view(np.int32)
    return out


def unpack_results(buf: np.ndarray) -> List[DecodeResult]:
    # bulk conversion to Python scalars once per rank: every rank may unpack EVERY rank's chunks
    if int(buf[0]) != _MAGIC:
        raise ValueError("result buffer: bad magic")
    n, ni, nd = int(buf[1]), int(buf[2]), int(buf[3])
    ints = buf[_HDR:_HDR + ni].tolist()
    dbl = np.ascontiguousarray(buf[_HDR + ni:_HDR + ni + 2 * nd]).view(np.float64).tolist()
    out, p, q = [], 0, 0
    for _ in range(n):
        k, kt, kf, kc = ints[p:p + 4]
        p += 4
        tok = tuple(ints[p:p + k]); p += k
        tim = frm = tc = None
        if kt >= 0:
            tim = ints[p:p + kt]; p += kt
        if kf >= 0:
            frm = ints[p:p + kf]; p += kf
        score, conf = dbl[q], dbl[q + 1]
        q += 2
        if kc >= 0:
            tc = dbl[q:q + kc]; q += kc
        r = DecodeResult(tok, score, confidence=conf, times=tim, tokens_confidence=tc)
        r.ctc_frames = frm
        out.append(r)
    return out


class GatheredResults(Sequence):
    """Every rank's per-chunk results in rank (= chunk) order, as they arrived: one packed buffer per rank.
    Behaves like a list of DecodeResult; a rank's rows become Python objects the first time one of them is read
    (each rank already holds its own chunks as DecodeResults -- turning the other ranks' rows into 240 Python
    scalars per chunk on EVERY rank costs about 1.5 ms per 176 chunks per source rank, which only a consumer that
    actually reads them should pay)."""

    def __init__(self, blocks):
        self._blocks = blocks                      # per rank: packed int32 buffer (host)
        self._starts = np.cumsum([0] + [int(b[1]) for b in blocks])
        self._cache = [None] * len(blocks)

    def __len__(self):
        return int(self._starts[-1])

    def _rank(self, r: int) -> List[DecodeResult]:
        if self._cache[r] is None:
            self._cache[r] = unpack_results(self._blocks[r])
        return self._cache[r]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        r = int(np.searchsorted(self._starts, i, side="right")) - 1
        return self._rank(r)[i - int(self._starts[r])]

    def __iter__(self):
        for r in range(len(self._blocks)):
            yield from self._rank(r)

    def total_tokens(self) -> int:
        return int(sum(int(b[4]) for b in self._blocks))


N_COLLECTIVES = 0        # all-gathers issued by all_gather_results so far (tests assert the common case is one per call)


class RvbComm:
    """The C-ABI collective (include/rvb.h: rvb_comm_unique_id / rvb_comm_create / rvb_comm_allgather): RCCL bound directly by
    librvb on a stream of its own, no torch tensor in the loop.  One communicator per process (= per GPU = per rank), shared
    by the ASR engine and the diarization engine of that rank.  The 128-byte id reaches the other ranks by a side channel:
    `from_torch_group` broadcasts it over an existing process group, `from_file` through a file on a shared directory
    (rank 0 writes it atomically, the others wait for it).  `engine` may be an Engine / DiarEngine (its device is used) or a
    device index."""

    def __init__(self, engine, world: int, rank: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        assert len(unique_id) == 128
        self.lib = _lib.load()
        self.world, self.rank = int(world), int(rank)
        device = _device_of(engine)
        self._id = C.create_string_buffer(unique_id, 128)
        self.handle = C.c_void_p()
        _lib.check(self.lib.rvb_comm_create(device, self.world, self.rank, C.cast(self._id, C.c_void_p), C.byref(self.handle)),
                   "rvb_comm_create")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().rvb_comm_unique_id(C.cast(buf, C.c_void_p)), "rvb_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_group(cls, engine):
        import torch.distributed as dist
        box = [cls.unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(engine, dist.get_world_size(), dist.get_rank(), box[0])

    @classmethod
    def from_file(cls, engine, world: int, rank: int, path: str, timeout: float = 120.0, token: str = None):
        """Rendezvous through a shared file.  The file carries a run token (default: MASTER_PORT / TORCHELASTIC_RUN_ID /
        RVB_RUN_ID, else "0") in front of the 128-byte id, so that a file left by an earlier run is never mistaken for
        this run's: rank 0 removes whatever is there before it writes, the others re-read until token and length match."""
        import os
        import time
        if token is None:
            token = os.environ.get("RVB_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
        head = ("rvbid:" + token + ":").encode()
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(head + cls.unique_id())
            os.replace(tmp, path)
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    blob = f.read()
            except FileNotFoundError:
                blob = b""
            if blob.startswith(head) and len(blob) == len(head) + 128:
                comm = cls(engine, world, rank, blob[len(head):])     # collective: returns once every rank has the id
                if rank == 0:
                    try:
                        os.unlink(path)                               # nothing stale is left for the next run
                    except OSError:
                        pass
                return comm
            if time.time() - t0 > timeout:
                raise TimeoutError(f"no RCCL id for run {token!r} at {path}")
            time.sleep(0.01)

    def all_gather(self, send: np.ndarray) -> np.ndarray:
        """send: C-contiguous array, equal size on every rank -> [world, ...] in rank order."""
        from ._lib import check
        send = np.ascontiguousarray(send)
        recv = np.empty((self.world,) + send.shape, send.dtype)
        check(self.lib.rvb_comm_allgather(self.handle, send.ctypes.data, send.nbytes, recv.ctypes.data), "rvb_comm_allgather")
        return recv

    def set_timeout(self, seconds: float):
        """Collectives that do not complete within `seconds` raise RvbError (RVB_E_TIMEOUT, -7) and kill the communicator
        instead of blocking for ever (0 = wait for ever)."""
        from ._lib import check
        check(self.lib.rvb_comm_set_timeout(self.handle, float(seconds)), "rvb_comm_set_timeout")

    def barrier(self):
        from ._lib import check
        check(self.lib.rvb_comm_barrier(self.handle), "rvb_comm_barrier")

    def max(self, value: float) -> float:
        """Maximum of `value` over the ranks."""
        import ctypes as C
        from ._lib import check
        v = C.c_double(float(value))
        check(self.lib.rvb_comm_max_f64(self.handle, C.byref(v)), "rvb_comm_max_f64")
        return float(v.value)

    def all_gather_topk(self, engine, to_host: bool = False):
        """The posterior exchange, device to device: every rank's [B, T, k] top-k CTC log-probs and ids of the engine's last
        encode, gathered from HBM into the communicator's device buffer.  -> (bytes per rank, host copy or None); the host copy
        is (vals [world, B, T, k] float32, ids [world, B, T, k] int32)."""
        import ctypes as C
        from ._lib import check
        n = C.c_int64(0)
        if not to_host:
            check(self.lib.rvb_comm_allgather_topk(self.handle, engine.handle, None, C.byref(n)), "rvb_comm_allgather_topk")
            return int(n.value), None
        v, i = engine.ctc_topk()                      # shapes only
        buf = np.empty(2 * self.world * v.size, np.int32)
        check(self.lib.rvb_comm_allgather_topk(self.handle, engine.handle, buf.ctypes.data, C.byref(n)), "rvb_comm_allgather_topk")
        half = self.world * v.size
        return int(n.value), (buf[:half].view(np.float32).reshape((self.world,) + v.shape), buf[half:].reshape((self.world,) + i.shape))

    def time_all_gather(self, nbytes: int, iters: int = 10) -> float:
        """Milliseconds of ONE all-gather of `nbytes` bytes per rank, device buffer to device buffer (HIP events on the
        communicator's stream, contents checked): the collective without the host staging all_gather() adds."""
        import ctypes as C
        from ._lib import check
        ms = C.c_double(0.0)
        check(self.lib.rvb_comm_time_allgather(self.handle, int(nbytes), int(iters), C.byref(ms)), "rvb_comm_time_allgather")
        return float(ms.value)

    def close(self):
        if self.handle:
            self.lib.rvb_comm_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_COMM = None


def default_comm(engine):
    """The transport of the result gathers: librvb's own RCCL binding (`rvb_comm_create` / `rvb_comm_allgather`,
    csrc/comm.hip -- no torch tensor in the loop) whenever the process group runs on GPUs ("nccl" = RCCL); None =
    torch.distributed (the gloo CPU tests with stub engines on a box without GPUs, or RVB_COMM=torch).  The choice depends on
    the backend, the presence of GPUs and RVB_COMM ONLY -- never on what this rank happens to hold (a rank without windows has no diarization engine yet: if it
    chose differently from its peers the collectives would not match and the job would hang).  `engine`: an Engine /
    DiarEngine, a device index, a torch device, or None (= LOCAL_RANK).  The unique id travels once through the existing
    process group; the communicator is process-wide."""
    import os
    import torch.distributed as dist
    global _DEFAULT_COMM
    if os.environ.get("RVB_COMM", "cabi") == "torch":
        return None
    if dist.get_backend() != "nccl":
        # a CPU process group: either the gloo tests with stub engines (torch.distributed does the gather, on CPU tensors), or
        # -- round 4 -- the rendezvous-only group of a launcher whose collectives are all librvb's.  The launcher says so
        # explicitly (bench.py / bench_diar.py export RVB_COMM=cabi before they create the group): a gloo group by itself, even
        # on a box with GPUs, is NOT taken as permission to bind RCCL (two stub ranks on one GPU would hang in ncclCommInitRank)
        import torch
        if os.environ.get("RVB_COMM") != "cabi" or not torch.cuda.is_available():
            return None
    if _COMM_FALLBACK is not None:                 # an earlier attempt of this process failed on some rank: torch.distributed
        return None
    if _DEFAULT_COMM is None or not _DEFAULT_COMM.handle:
        _DEFAULT_COMM = _create_comm_or_agree_on_fallback(_device_of(engine))
    return _DEFAULT_COMM


_COMM_FALLBACK = None      # why librvb's RCCL communicator is not used by this process (None = it is, or was never tried)
_COMM_INIT_HUNG = False    # a thread of this process is still inside ncclCommInitRank (the caller should leave with os._exit)


def comm_fallback_reason():
    """None, or the reason every rank of this job exchanges through torch.distributed instead of librvb's RCCL binding."""
    return _COMM_FALLBACK


def comm_init_hung() -> bool:
    return _COMM_INIT_HUNG


def _create_comm_or_agree_on_fallback(device: int):
    """Create the process-wide RvbComm, or -- when that fails on ANY rank -- make every rank fall back to torch.distributed
    together (`ncclCommInitRank` across processes first runs on the driver's scaling box, and a failure
    there used to raise and kill `bench.py --gpus 8`).  Steps, in lockstep on every rank: (1) the id travels over the torch
    group and rvb_comm_create runs in a helper thread with a deadline (RVB_COMM_INIT_TIMEOUT, default 120 s: an init that
    hangs counts as failed); (2) preflight: one 1-KB all-gather (contents checked) under a 30 s collective timeout, so that the
    first collective of the timed region cannot be the first ever; (3) the ranks all-gather (ok, reason) over the torch
    group; unless every rank is fine, every rank frees its communicator, records the reason (comm_fallback_reason), logs it
    to stderr and returns None -- the callers then gather through torch.distributed (the rendezvous group: gloo on host
    buffers, or torch's own RCCL communicator)."""
    import os
    import sys
    import threading
    import torch.distributed as dist
    global _COMM_FALLBACK, _COMM_INIT_HUNG
    comm, reason = None, None
    box = {}

    if dist.get_world_size() == 1:
        return RvbComm.from_torch_group(device)      # one rank: nothing to wait for, nothing to agree on
    ident = [RvbComm.unique_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(ident, src=0)         # (torch collectives stay on the calling thread: its current device)

    def create():
        try:
            box["comm"] = RvbComm(device, dist.get_world_size(), dist.get_rank(), ident[0])
        except BaseException as ex:          # noqa: BLE001 -- whatever it is, the ranks must hear about it
            box["error"] = f"{type(ex).__name__}: {ex}"

    t = threading.Thread(target=create, name="rvb_comm_create", daemon=True)
    t.start()
    t.join(float(os.environ.get("RVB_COMM_INIT_TIMEOUT", "120")))
    if t.is_alive():
        _COMM_INIT_HUNG = True
        reason = "rvb_comm_create did not return within RVB_COMM_INIT_TIMEOUT"
    elif "error" in box:
        reason = box["error"]
    else:
        comm = box["comm"]
        try:
            try:
                comm.set_timeout(30.0)
            except Exception:                 # an RCCL build without ncclCommAbort refuses timeouts: preflight without one
                pass
            got = comm.all_gather(np.full(256, comm.rank, np.int32))      # through comm_wait: honours the timeout (the timed form does not)
            if [int(r[0]) for r in got] != list(range(comm.world)):
                raise RuntimeError("the slots do not hold their ranks' words")
            try:
                comm.set_timeout(0.0)
            except Exception:
                pass
        except Exception as ex:
            reason = f"preflight all-gather failed: {type(ex).__name__}: {ex}"
    votes = [None] * dist.get_world_size()
    dist.all_gather_object(votes, reason)
    bad = [(r, v) for r, v in enumerate(votes) if v is not None]
    if not bad:
        return comm
    if comm is not None:
        try:
            comm.close()
        except Exception:
            pass
    _COMM_FALLBACK = "; ".join(f"rank {r}: {v}" for r, v in bad)
    sys.stderr.write("reverb_amd.dist: librvb's RCCL communicator is not usable (" + _COMM_FALLBACK +
                     "); every rank exchanges through torch.distributed (" + dist.get_backend() + ") instead\n")
    sys.stderr.flush()
    return None


def _device_of(engine) -> int:
    """Device index of an engine / int / torch.device / "cuda:N" string; LOCAL_RANK when nothing says otherwise."""
    import os
    if engine is None:
        return int(os.environ.get("LOCAL_RANK", "0"))
    if isinstance(engine, int):
        return engine
    if isinstance(engine, str):
        return int(engine.split(":")[1]) if ":" in engine else int(os.environ.get("LOCAL_RANK", "0"))
    idx = getattr(engine, "index", None)            # torch.device
    if idx is None:
        idx = getattr(engine, "device_index", None)
    if idx is None:
        idx = getattr(engine, "device", None)
        if not isinstance(idx, int):
            idx = getattr(idx, "index", None)
    return int(idx) if idx is not None else int(os.environ.get("LOCAL_RANK", "0"))


def gather_words(send: np.ndarray, device, comm: "RvbComm" = None) -> np.ndarray:
    """ONE all-gather of equal-size int32 buffers: -> [world, len(send)] on the host, in rank order."""
    global N_COLLECTIVES
    send = np.ascontiguousarray(send, dtype=np.int32)
    N_COLLECTIVES += 1
    if comm is not None:                           # librvb's own RCCL binding (rvb_allgather_results)
        return comm.all_gather(send)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if dist.get_backend() != "nccl":
        device = "cpu"                              # gloo gathers host tensors
    t = torch.from_numpy(send).to(device)
    g = torch.empty(world * send.size, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(g, t)
    return g.cpu().numpy().reshape(world, send.size)   # one device-to-host copy


def all_gather_results(hyps: Sequence[DecodeResult], device, max_count: int = None, comm: "RvbComm" = None) -> "GatheredResults":
    """All ranks end up with every rank's results in rank (= chunk) order after ONE all-gather.  Payload: tokens,
    CTC peak frames, score, confidences -- about 1.3 KB per chunk, latency-bound on xGMI.
    `max_count` = the largest number of results any rank contributes; it sizes the fixed-capacity buffer and must
    be the same on every rank (default: len(hyps), i.e. every rank holds equally many).  `comm`: the RvbComm of
    default_comm() -- librvb's C-ABI collective -- or None for torch.distributed."""
    import torch.distributed as dist
    global _ROW_WORDS
    world = comm.world if comm is not None else dist.get_world_size()
    buf = pack_results(hyps)
    count = max(int(max_count if max_count is not None else len(hyps)), 1)
    while True:
        cap = _HDR + count * _ROW_WORDS
        send = np.zeros(cap, np.int32)
        m = min(cap, buf.size)
        send[:m] = buf[:m]                         # the header always fits and states what this rank needs
        host = gather_words(send, device, comm)
        need = int((_HDR + host[:, 2].astype(np.int64) + 2 * host[:, 3].astype(np.int64)).max())
        if need <= cap:
            break
        _ROW_WORDS = -(-(need - _HDR) // count) * 5 // 4 + 8       # same arithmetic on the same numbers on every rank
    return GatheredResults([host[r, :_HDR + int(host[r, 2]) + 2 * int(host[r, 3])] for r in range(world)])


def share_fp8_scales(engine, n_frames: int, chunk_size: int, beam_size: int, device) -> None:
    """fp8 engines calibrate their activation scales on the first batch they encode; the ranks of a sharded run would each
    do so on their own slice and quantise the same recording differently.  Here every rank that is not calibrated
    yet runs one calibration pass over its slice, the scales (powers of two) are all-gathered, and every rank installs the
    element-wise maximum -- what a single GPU calibrating on the whole recording measures.  No-op for other engines and for
    engines that are calibrated already (scales stick to an engine across recordings)."""
    if getattr(engine, "dtype", None) != "fp8" or not hasattr(engine, "fp8_scales"):
        return
    has_vec = hasattr(engine, "fp8_scale_vector")        # with conv1's output scale; stubs: blocks only
    vector = engine.fp8_scale_vector if has_vec else engine.fp8_scales
    mine = vector()
    if mine is None and n_frames > 0:
        engine.decode_resident(n_frames, ["ctc_greedy_search"], chunk_size, beam_size, 0.0, 0.0)      # bf16 pass, records max |.|
        mine = vector()
    nb = int(engine.cfg.num_blocks)
    send = np.zeros(1 + nb * 7 + (1 if has_vec else 0), np.float32)      # (`vector is not engine.fp8_scales` was always true)
    if mine is not None:
        send[0] = 1.0
        send[1:] = mine.reshape(-1)
    host = gather_words(send.view(np.int32), device, default_comm(engine)).view(np.float32)
    have = host[:, 0] > 0
    if have.any():
        engine.set_fp8_scales(host[have, 1:].max(axis=0))


def share_emb_fp8_scales(pipeline, device) -> bool:
    """The diarization counterpart of share_fp8_scales: the embedding trunk of an fp8 pipeline calibrates its 32 activation
    scales on the first windows it sees (that pass runs in bf16).  On the first sharded recording of a pipeline every rank
    reports the scales it holds (none: a rank without windows has no engine), the element-wise maximum is installed
    everywhere (rvd_set_emb_fp8_scales; kept on the pipeline for engines created later) and True is returned -- the caller
    embeds its windows again, so ONE quantisation feeds the clustering whatever the world size.  Whether the collective
    runs depends on the pipeline's dtype and on how many sharded calls it has seen ONLY -- the same on every rank."""
    if getattr(pipeline, "dtype", None) != "fp8" or getattr(pipeline, "_emb_fp8_scales", None) is not None:
        return False
    eng = getattr(pipeline, "_engine", None)
    send = np.zeros(33, np.float32)
    if eng is not None and hasattr(eng, "emb_fp8"):
        state, scales, _ = eng.emb_fp8()
        if state == 2:
            send[0] = 1.0
            send[1:] = scales
    idx = getattr(pipeline, "device_index", None)
    host = gather_words(send.view(np.int32), device, default_comm(idx if idx is not None else device)).view(np.float32)
    have = host[:, 0] > 0
    if not have.any():
        return False              # nobody embedded anything (no active speaker anywhere): try again on the next recording
    pipeline._emb_fp8_scales = host[have, 1:].max(axis=0).astype(np.float32)
    if eng is not None and hasattr(eng, "set_emb_fp8_scales"):
        eng.set_emb_fp8_scales(pipeline._emb_fp8_scales)
    return True


class CollectiveFailed(RuntimeError):
    """The result gather did not complete (a peer died or hangs); decode_sharded switches to the store exchange."""


_EPOCH = 0          # decode_sharded calls so far: every rank calls in lockstep, so this names one call's keys in the store
_KNOWN_DEAD: set = set()    # ranks a recovery declared dead: later calls give them no units and do not wait for them


def _drop_default_comm():
    """After a failed collective the RCCL communicator is aborted and refuses every further call (rvb_comm_set_timeout):
    free it and forget it, so that nothing later in the process (another recording, the caller's own barrier) is handed a
    dead handle.  With dead ranks on record the sharded entry points exchange through the store from then on;
    with none (a transient failure) the next call builds a fresh communicator."""
    global _DEFAULT_COMM
    if _DEFAULT_COMM is not None:
        try:
            _DEFAULT_COMM.close()
        except Exception:
            pass
        _DEFAULT_COMM = None


def ranges_over_alive(n_units: int, world: int) -> List[Tuple[int, int]]:
    """chunk_ranges over the ranks not known to be dead; a dead rank keeps an empty range at the position where its units
    would have started, so the list stays in unit order and `ranges[rank]` stays valid for every rank."""
    alive = [r for r in range(world) if r not in _KNOWN_DEAD]
    parts = chunk_ranges(n_units, max(len(alive), 1))
    out, c, k = [], 0, 0
    for r in range(world):
        if r in _KNOWN_DEAD or not alive:
            out.append((c, c))
        else:
            out.append(parts[k]); c = parts[k][1]; k += 1
    return out


def _store():
    """The key-value store the process group was built on (TCPStore: hosted by rank 0 or by the launcher's agent).  It is the
    side channel of the failure path: it keeps working when a collective cannot complete."""
    from torch.distributed.distributed_c10d import _get_default_store
    return _get_default_store()


def _store_wait(store, key: str, seconds: float):
    """bytes of `key`, or None if it does not appear within `seconds`."""
    import datetime
    try:
        store.wait([key], datetime.timedelta(seconds=max(seconds, 0.001)))
        return store.get(key)
    except Exception:           # DistStoreError / RuntimeError("... timed out"), depending on the torch version
        return None


def _decode_range(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, c0, c1):
    s0, s1 = sample_range(len(pcm), chunk_size, c0, c1)
    if s1 <= s0:
        return {m: [] for m in modes}, 0
    engine.upload_pcm(pcm[s0:s1])
    nf = engine.fbank()
    return engine.decode_resident(nf, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty), nf


def _recover_through_store(what: str, ranges, local_blob: bytes, redo, timeout: float, why: str):
    """The failure path shared by decode_sharded and diarize_sharded ("worker failure -> re-queue chunk
    range"; the reference is a single process and has nothing of the kind).  The units of both paths (chunks, windows) are
    independent and idempotent (asr/wenet/cli/reverb.py:220-253), so a missing rank costs only its range:

      1. every surviving rank puts its packed results into the rendezvous store (`rvb/<epoch>/res/<rank>`);
      2. rank 0 waits `timeout` seconds for them, declares the ranks whose results did not arrive dead, splits their ranges
         evenly over the survivors and publishes that plan (`rvb/<epoch>/plan`);
      3. every survivor computes the extra ranges the plan gives it (`redo(a, b) -> bytes`) and publishes them
         (`rvb/<epoch>/extra/<i>`);
      4. every survivor reads all pieces.

    Rank 0 is the arbiter (it usually hosts the store as well): if rank 0 itself is lost the job fails -- stated, not hidden.
    Returns (info, pieces): info = {"alive", "dead", "plan", "why"}, pieces = [(a, b, blob)] covering every range in order."""
    import json
    import time
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    store = _store()
    tag = f"rvb/{_EPOCH}"
    _drop_default_comm()
    store.set(f"{tag}/res/{rank}", local_blob)
    if rank == 0:
        # ONE deadline for all ranks (not `timeout` per rank: with several dead ranks the survivors would give up on the plan
        # first); ranks already on record as dead are not waited for at all
        deadline = time.time() + timeout
        blobs = {r: None if r in _KNOWN_DEAD else _store_wait(store, f"{tag}/res/{r}", deadline - time.time()) for r in range(world)}
        alive = [r for r in range(world) if blobs[r] is not None]
        dead = [r for r in range(world) if blobs[r] is None]
        plan = []                                   # [dead rank, survivor, a, b] in unit order
        for d in dead:
            a, b = ranges[d]
            for surv, (u, v) in zip(alive, chunk_ranges(b - a, len(alive))):
                if v > u:
                    plan.append([d, surv, a + u, a + v])
        store.set(f"{tag}/plan", json.dumps({"alive": alive, "dead": dead, "plan": plan, "why": why}).encode())
    long_wait = 3.0 * timeout + 30.0
    blob = _store_wait(store, f"{tag}/plan", long_wait)
    if blob is None:
        raise CollectiveFailed(f"{what}: no recovery plan from rank 0 within {long_wait:.0f} s ({why})")
    info = json.loads(bytes(blob).decode())
    _KNOWN_DEAD.update(info["dead"])
    if rank not in info["alive"]:
        raise CollectiveFailed(f"{what}: rank {rank} was declared dead by rank 0 (its results arrived after {timeout} s)")
    for i, (d, surv, a, b) in enumerate(info["plan"]):
        if surv == rank:                            # the re-queued work: a dead rank's units, computed here
            store.set(f"{tag}/extra/{i}", redo(a, b))
    pieces = []
    for r, (a, b) in enumerate(ranges):
        if r in info["alive"]:
            piece = _store_wait(store, f"{tag}/res/{r}", timeout)
            if piece is None:
                raise CollectiveFailed(f"{what}: results of rank {r} vanished from the store")
            pieces.append((a, b, bytes(piece)))
        else:
            for i, (d, surv, u, v) in enumerate(info["plan"]):
                if d != r:
                    continue
                piece = _store_wait(store, f"{tag}/extra/{i}", long_wait)
                if piece is None:
                    raise CollectiveFailed(f"{what}: re-queued range [{u}, {v}) of dead rank {r} never arrived from rank {surv}")
                pieces.append((u, v, bytes(piece)))
    return info, pieces


def _recover_sharded(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, local, ranges, timeout,
                     why: str):
    """decode_sharded after a failed gather: -> {mode: results of all chunks} like the fast path (_recover_through_store)."""
    nm = len(modes)

    def pack(res):
        return pack_results([h for m in modes for h in res[m]]).tobytes()

    def redo(c0, c1):
        return pack(_decode_range(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, c0, c1)[0])

    info, pieces = _recover_through_store("decode_sharded", ranges, pack(local), redo, timeout, why)
    out = {m: [] for m in modes}
    for a, b, blob in pieces:
        rows = unpack_results(np.frombuffer(blob, np.int32))
        assert len(rows) == (b - a) * nm, (len(rows), a, b, nm)
        for i, m in enumerate(modes):
            out[m].extend(rows[i * (b - a):(i + 1) * (b - a)])
    decode_sharded.last_recovery = info            # what happened, for the caller's log
    return out


def decode_sharded(engine, pcm: np.ndarray, modes, chunk_size: int, beam_size: int, ctc_weight: float,
                   reverse_weight: float, device, blank_penalty: float = 0.0, timeout: float = None):
    """Decode one long recording with every rank of the default process group taking a contiguous
    chunk range; returns {mode: results of ALL chunks} on every rank.

    `timeout` (seconds; None = wait for ever, as round 3 did): how long the result gather may take before a peer is presumed
    dead.  The RCCL collective then returns RVB_E_TIMEOUT instead of hanging (rvb_comm_set_timeout), a gloo / torch
    collective raises after the process group's own timeout; either way the ranks that are still there exchange through the
    rendezvous store and re-decode the missing rank's chunk range (_recover_sharded)."""
    import torch.distributed as dist
    global _EPOCH
    _EPOCH += 1
    decode_sharded.last_recovery = None
    world, rank = dist.get_world_size(), dist.get_rank()
    n_chunks = -(-num_frames(len(pcm)) // chunk_size)
    ranges = ranges_over_alive(n_chunks, world)
    c0, c1 = ranges[rank]
    s0, s1 = sample_range(len(pcm), chunk_size, c0, c1)
    local = {m: [] for m in modes}
    nf = 0
    if s1 > s0:
        engine.upload_pcm(pcm[s0:s1])
        nf = engine.fbank()
    if _KNOWN_DEAD:
        # an earlier call lost ranks: no collective can complete any more.  The survivors share the recording among
        # themselves and exchange through the store straight away (nobody waits for the dead; a rank that dies NOW is
        # found and its range re-queued as before)
        if s1 > s0:
            local = engine.decode_resident(nf, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty)
        return _recover_sharded(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, local, ranges,
                                float(timeout if timeout is not None else 60.0), "ranks %s lost earlier" % sorted(_KNOWN_DEAD))
    comm = default_comm(engine)
    if comm is not None and timeout is not None:
        comm.set_timeout(timeout)
    try:
        share_fp8_scales(engine, nf, chunk_size, beam_size, device)
        if s1 > s0:
            local = engine.decode_resident(nf, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty)
        # every mode's rows travel in the same single all-gather: rank block = [mode 0 rows | mode 1 rows | ...]
        kmax = max(b - a for a, b in ranges)
        merged = all_gather_results([h for m in modes for h in local[m]], device, max_count=kmax * len(modes), comm=comm)
    except Exception as ex:
        # a collective that cannot complete: RvbError (RVB_E_TIMEOUT / a dead communicator), or whatever the torch backend
        # raises when a peer is gone.  Without a timeout the caller asked for round 3's behaviour: no recovery.
        if timeout is None or not _is_collective_failure(ex):
            raise
        if s1 > s0 and not any(local[m] for m in modes):       # the failure came before this rank's own decode (fp8 scale exchange)
            local, _ = _decode_range(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, c0, c1)
        return _recover_sharded(engine, pcm, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty, local, ranges,
                                float(timeout), f"{type(ex).__name__}: {ex}")
    out = {m: [] for m in modes}
    for r, (a, b) in enumerate(ranges):
        rows = merged._rank(r)
        assert len(rows) == (b - a) * len(modes), (len(rows), a, b)
        for i, m in enumerate(modes):
            out[m].extend(rows[i * (b - a):(i + 1) * (b - a)])
    return out


decode_sharded.last_recovery = None


def _is_collective_failure(ex: Exception) -> bool:
    from ._lib import RvbError
    if isinstance(ex, RvbError):
        return "(-7)" in str(ex) or "aborted after a timeout" in str(ex)
    if isinstance(ex, (AssertionError, KeyboardInterrupt)):
        return False
    text = str(ex).lower()
    return isinstance(ex, RuntimeError) and any(w in text for w in ("timed out", "timeout", "connection", "peer", "nccl", "gloo", "socket"))


# ------------------------------------------------------------------------------------------------ diarization
def window_sample_range(n_samples: int, window: int, step: int, w0: int, w1: int) -> Tuple[int, int]:
    """Samples a rank needs for diarization windows [w0, w1): window w covers [w*step, w*step + window); the
    slice ends where the last window ends, or at the end of the file (the zero-padded tail window)."""
    if w1 <= w0:
        return 0, 0
    return w0 * step, min(n_samples, (w1 - 1) * step + window)


def segmentation_frames(window_samples: int) -> int:
    """Output frames of the PyanNet segmentation model for one window: SincNet (conv k=251 stride 10, then three times
    {max-pool 3, conv k=5} with the pool in front of the last two convolutions) -- 160 000 samples -> 589 frames."""
    n = (window_samples - 251) // 10 + 1
    n = n // 3
    n = (n - 5 + 1) // 3
    n = (n - 5 + 1) // 3
    return n


def pack_diar_shard(classes: Optional[np.ndarray], emb: Optional[np.ndarray], kmax: int, frames: int, dim: int) -> np.ndarray:
    """One rank's windows as int32 words: [n_windows, frames, dim, 0 | classes uint8 (kmax x frames, padded to words) |
    embeddings fp32 bit patterns (kmax x 3 x dim)]; fixed size for given (kmax, frames, dim)."""
    cw = -(-(kmax * frames) // 4)
    out = np.zeros(4 + cw + kmax * 3 * dim, np.int32)
    n = 0 if classes is None else classes.shape[0]
    out[:4] = (n, frames, dim, 0)
    if n:
        assert classes.shape == (n, frames) and emb.shape == (n, 3, dim) and n <= kmax
        cb = np.zeros(cw * 4, np.uint8)
        cb[:n * frames] = np.ascontiguousarray(classes, np.uint8).reshape(-1)
        out[4:4 + cw] = cb.view(np.int32)
        e = np.full((kmax, 3, dim), np.nan, np.float32)
        e[:n] = emb
        out[4 + cw:] = e.reshape(-1).view(np.int32)
    return out


def unpack_diar_shard(words: np.ndarray, kmax: int):
    n, frames, dim = int(words[0]), int(words[1]), int(words[2])
    cw = -(-(kmax * frames) // 4)
    classes = words[4:4 + cw].view(np.uint8)[:n * frames].reshape(n, frames).copy()
    emb = words[4 + cw:4 + cw + kmax * 3 * dim].view(np.float32).reshape(kmax, 3, dim)[:n].copy()
    return classes, emb


def diarize_sharded(pipeline, pcm: np.ndarray, device, uri=None, timeout: float = None, **kwargs):
    """Diarize one long recording with the 10 s windows of pyannote's sliding inference split into contiguous
    ranges, one per rank (one process per GPU).  Both networks run on the rank's own windows with no data-path
    collective; ONE all-gather (the same transport as the ASR results: librvb's rvb_allgather_results on GPUs) then
    brings the per-window powerset classes (uint8, 589 B per window) and the speaker embeddings (3 x 256 fp32 per
    window) of every rank to every rank in one packed buffer, and the global part -- speaker count, clustering,
    reconstruction -- runs identically everywhere.  Returns the Annotation on every rank.

    `timeout` (seconds; None = wait for ever): as in decode_sharded -- a gather that cannot complete is replaced by the
    exchange through the rendezvous store, and the windows of a rank that is gone are run by the survivors (round 4)."""
    import torch.distributed as dist
    global _EPOCH
    _EPOCH += 1
    diarize_sharded.last_recovery = None
    world, rank = dist.get_world_size(), dist.get_rank()
    cfg = pipeline.cfg
    win, step = int(cfg["window_samples"]), int(cfg["step_samples"])
    n = len(pcm)
    full = (n - win) // step + 1 if n >= win else 0
    n_windows = full + (1 if (n < win or (n - win) % step > 0) else 0)
    ranges = ranges_over_alive(n_windows, world)
    w0, w1 = ranges[rank]
    frames = segmentation_frames(win)               # known without running the network: ranks with no window need it too
    dim = int(cfg["emb_dim"])

    def run_windows(a, b):
        if b <= a:
            return None, None
        s0, s1 = window_sample_range(n, win, step, a, b)
        classes, emb = pipeline.networks(pcm[s0:s1], prepare_finish=False)    # finish() sees the gathered windows, not this slice
        assert classes.shape == (b - a, frames), (classes.shape, a, b, frames)
        return classes, emb

    classes, emb = run_windows(w0, w1)
    if not _KNOWN_DEAD and share_emb_fp8_scales(pipeline, device):
        # fp8 trunk, first recording of these engines: the pass above calibrated each rank on its own windows (and ran its first
        # trunk pass in bf16); now that every rank holds the same scales, embed the windows again so that ONE quantisation feeds
        # the clustering whatever the world size.  Once per engine: calibrated engines skip both steps.
        classes, emb = run_windows(w0, w1)
    kmax = max(b - a for a, b in ranges)
    comm = None
    if not _KNOWN_DEAD:
        comm = default_comm(pipeline.device_index if getattr(pipeline, "device_index", None) is not None else device)
        if comm is not None and timeout is not None:
            comm.set_timeout(timeout)
    try:
        if _KNOWN_DEAD:                             # ranks were lost by an earlier call: straight to the store exchange
            if timeout is None:
                timeout = 60.0
            raise CollectiveFailed("ranks %s lost earlier" % sorted(_KNOWN_DEAD))
        host = gather_words(pack_diar_shard(classes, emb, kmax, frames, dim), device, comm)
        parts = [unpack_diar_shard(host[r], kmax) for r in range(world)]
        for r, (a, b) in enumerate(ranges):
            assert parts[r][0].shape[0] == b - a, (r, parts[r][0].shape, a, b)
    except Exception as ex:
        if timeout is None or not (isinstance(ex, CollectiveFailed) or _is_collective_failure(ex)):
            raise

        def redo(a, b):
            c, e = run_windows(a, b)
            return pack_diar_shard(c, e, b - a, frames, dim).tobytes()

        info, pieces = _recover_through_store("diarize_sharded", ranges, pack_diar_shard(classes, emb, max(w1 - w0, 1), frames, dim).tobytes(),
                                              redo, float(timeout), f"{type(ex).__name__}: {ex}")
        parts = []
        for a, b, blob in pieces:
            c, e = unpack_diar_shard(np.frombuffer(blob, np.int32), max(b - a, 1))
            assert c.shape[0] == b - a, (c.shape, a, b)
            parts.append((c, e))
        diarize_sharded.last_recovery = info
    all_c = np.concatenate([c for c, _ in parts])
    all_e = np.concatenate([e for _, e in parts])
    return pipeline.finish(all_c, all_e, uri, **kwargs)


diarize_sharded.last_recovery = None
