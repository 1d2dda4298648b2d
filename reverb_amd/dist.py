"""Chunk-sharded multi-GPU decoding: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), no data-path collective, one all-gather of the per-chunk results at the end.

The reference has no multi-GPU inference; what makes the shard exact is that its
long-form driver carries no state between chunks: `feats_batcher` cuts fixed, non-overlapping
chunks and each batch is decoded on its own (asr/wenet/cli/reverb.py:148-180, 220-253); only the
per-chunk time offset (`:320-325`) couples them, on the host.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import List, Tuple

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


def pack_results(hyps: Sequence[DecodeResult], lmax: int):
    n = len(hyps)
    ints = np.full((n, 2 * lmax + 2), -1, np.int32)
    flts = np.zeros((n, lmax + 2), np.float64)
    for i, h in enumerate(hyps):
        k, kt = len(h.tokens), len(h.times or [])
        ints[i, 0], ints[i, 1] = k, kt
        ints[i, 2:2 + k] = h.tokens
        if kt:
            ints[i, 2 + lmax:2 + lmax + kt] = h.times
        flts[i, 0], flts[i, 1] = h.score, h.confidence
        if h.tokens_confidence:
            flts[i, 2:2 + k] = h.tokens_confidence
    return ints, flts


def unpack_results(ints: np.ndarray, flts: np.ndarray, n: int, lmax: int) -> List[DecodeResult]:
    # one bulk conversion to Python scalars per array: every rank unpacks EVERY rank's chunks, so this runs
    # world x chunks times per call (element-wise int()/float() cost 28 ms per step at 8 x 176 chunks)
    rows_i, rows_f = ints[:n].tolist(), flts[:n].tolist()
    out = []
    for ri, rf in zip(rows_i, rows_f):
        k, kt = ri[0], ri[1]
        out.append(DecodeResult(tuple(ri[2:2 + k]), rf[0], confidence=rf[1], times=ri[2 + lmax:2 + lmax + kt],
                                tokens_confidence=rf[2:2 + k]))
    return out


class GatheredResults(Sequence):
    """Every rank's per-chunk results in rank (= chunk) order, as they arrived: packed integer / float rows per rank.
    Behaves like a list of DecodeResult; a rank's rows become Python objects the first time one of them is read
    (each rank already holds its own chunks as DecodeResults -- turning the other ranks' rows into 240 Python
    scalars per chunk on EVERY rank costs about 1.5 ms per 176 chunks per source rank, which only a consumer that
    actually reads them should pay)."""

    def __init__(self, blocks, lmax: int):
        self._blocks = blocks                      # [(ints, flts, count)] per rank, host arrays
        self._lmax = lmax
        self._starts = np.cumsum([0] + [b[2] for b in blocks])
        self._cache = [None] * len(blocks)

    def __len__(self):
        return int(self._starts[-1])

    def _rank(self, r: int) -> List[DecodeResult]:
        if self._cache[r] is None:
            ints, flts, n = self._blocks[r]
            self._cache[r] = unpack_results(ints, flts, n, self._lmax)
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
        return int(sum(int(b[0][:b[2], 0].sum()) for b in self._blocks))


def all_gather_results(hyps: Sequence[DecodeResult], device) -> "GatheredResults":
    """All ranks end up with every rank's results in rank (= chunk) order.  Payload: tokens, CTC peak
    frames, score, confidences -- about 1 KB per chunk, latency-bound on xGMI."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    lmax_local = max([len(h.tokens) for h in hyps] + [len(h.times or []) for h in hyps] + [1])
    meta = torch.tensor([len(hyps), lmax_local], device=device, dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0]) for m in metas]
    nmax, lmax = max(counts + [1]), max(int(m[1]) for m in metas)
    ints, flts = pack_results(hyps, lmax)
    pad_i = np.full((nmax, ints.shape[1]), -1, np.int32)
    pad_f = np.zeros((nmax, flts.shape[1]), np.float64)
    pad_i[:len(hyps)], pad_f[:len(hyps)] = ints, flts
    ti, tf = torch.from_numpy(pad_i).to(device), torch.from_numpy(pad_f).to(device)
    gi = [torch.empty_like(ti) for _ in range(world)]
    gf = [torch.empty_like(tf) for _ in range(world)]
    dist.all_gather(gi, ti)
    dist.all_gather(gf, tf)
    hi, hf = torch.stack(gi).cpu().numpy(), torch.stack(gf).cpu().numpy()    # two device-to-host copies, not 2 x world
    return GatheredResults([(hi[r], hf[r], counts[r]) for r in range(world)], lmax)


def decode_sharded(engine, pcm: np.ndarray, modes, chunk_size: int, beam_size: int, ctc_weight: float,
                   reverse_weight: float, device, blank_penalty: float = 0.0):
    """Decode one long recording with every rank of the default process group taking a contiguous
    chunk range; returns {mode: results of ALL chunks} on every rank."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    n_chunks = -(-num_frames(len(pcm)) // chunk_size)
    c0, c1 = chunk_ranges(n_chunks, world)[rank]
    s0, s1 = sample_range(len(pcm), chunk_size, c0, c1)
    local = {m: [] for m in modes}
    if s1 > s0:
        engine.upload_pcm(pcm[s0:s1])
        nf = engine.fbank()
        local = engine.decode_resident(nf, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty)
    return {m: all_gather_results(local[m], device) for m in modes}


# ------------------------------------------------------------------------------------------------ diarization
def window_sample_range(n_samples: int, window: int, step: int, w0: int, w1: int) -> Tuple[int, int]:
    """Samples a rank needs for diarization windows [w0, w1): window w covers [w*step, w*step + window); the
    slice ends where the last window ends, or at the end of the file (the zero-padded tail window)."""
    if w1 <= w0:
        return 0, 0
    return w0 * step, min(n_samples, (w1 - 1) * step + window)


def diarize_sharded(pipeline, pcm: np.ndarray, device, uri=None, **kwargs):
    """Diarize one long recording with the 10 s windows of pyannote's sliding inference split into contiguous
    ranges, one per rank (one process per GPU).  Both networks run on the rank's own windows with no data-path
    collective; one all-gather then brings the per-window powerset classes (uint8, 589 B per window) and the
    speaker embeddings (3 x 256 fp32 per window) to every rank, and the global part -- speaker count,
    clustering, reconstruction -- runs identically everywhere.  Returns the Annotation on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    cfg = pipeline.cfg
    win, step = int(cfg["window_samples"]), int(cfg["step_samples"])
    n = len(pcm)
    full = (n - win) // step + 1 if n >= win else 0
    n_windows = full + (1 if (n < win or (n - win) % step > 0) else 0)
    ranges = chunk_ranges(n_windows, world)
    w0, w1 = ranges[rank]
    frames = None
    if w1 > w0:
        s0, s1 = window_sample_range(n, win, step, w0, w1)
        classes, emb = pipeline.networks(pcm[s0:s1])
        assert classes.shape[0] == w1 - w0, (classes.shape, w0, w1)
        frames = classes.shape[1]
    fr = torch.tensor([frames or 0], device=device, dtype=torch.int64)
    dist.all_reduce(fr, op=dist.ReduceOp.MAX)
    frames = int(fr.item())
    dim = int(cfg["emb_dim"])
    kmax = max(b - a for a, b in ranges)
    pc = np.zeros((kmax, frames), np.uint8)
    pe = np.full((kmax, 3, dim), np.nan, np.float32)
    if w1 > w0:
        pc[:w1 - w0], pe[:w1 - w0] = classes, emb
    tc, te = torch.from_numpy(pc).to(device), torch.from_numpy(pe).to(device)
    gc = [torch.empty_like(tc) for _ in range(world)]
    ge = [torch.empty_like(te) for _ in range(world)]
    dist.all_gather(gc, tc)
    dist.all_gather(ge, te)
    all_c = np.concatenate([gc[r].cpu().numpy()[:b - a] for r, (a, b) in enumerate(ranges)])
    all_e = np.concatenate([ge[r].cpu().numpy()[:b - a] for r, (a, b) in enumerate(ranges)])
    return pipeline.finish(all_c, all_e, uri, **kwargs)
