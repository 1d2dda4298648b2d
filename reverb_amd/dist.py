"""Chunk-sharded multi-GPU decoding: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), no data-path collective, one all-gather of the per-chunk results at the end.

The reference has no multi-GPU inference; what makes the shard exact is that its
long-form driver carries no state between chunks: `feats_batcher` cuts fixed, non-overlapping
chunks and each batch is decoded on its own (asr/wenet/cli/reverb.py:148-180, 220-253); only the
per-chunk time offset (`:320-325`) couples them, on the host.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

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
    out = []
    for i in range(n):
        k, kt = int(ints[i, 0]), int(ints[i, 1])
        out.append(DecodeResult(tuple(int(t) for t in ints[i, 2:2 + k]), float(flts[i, 0]), confidence=float(flts[i, 1]),
                                times=[int(t) for t in ints[i, 2 + lmax:2 + lmax + kt]],
                                tokens_confidence=[float(c) for c in flts[i, 2:2 + k]]))
    return out


def all_gather_results(hyps: Sequence[DecodeResult], device) -> List[DecodeResult]:
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
    merged: List[DecodeResult] = []
    for r in range(world):
        merged.extend(unpack_results(gi[r].cpu().numpy(), gf[r].cpu().numpy(), counts[r], lmax))
    return merged


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
