"""Multi-GPU path on CPU: the chunk partition, the waveform slicing with its 240-sample halo, and
the result all-gather (world_size 2, gloo backend -- the same torch.distributed calls run over
RCCL on the GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import fbank_ref
from reverb_amd import dist as rdist
from reverb_amd import synth
from reverb_amd.search import DecodeResult


def test_chunk_ranges_cover_everything_in_order():
    for n in (0, 1, 7, 8, 176, 1405):
        for w in (1, 2, 4, 8):
            r = rdist.chunk_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert rdist.chunk_ranges(1405, 8)[0] == (0, 176) and rdist.chunk_ranges(1405, 8)[7] == (1230, 1405)


def test_sample_slices_reproduce_the_whole_file_features():
    """fbank of each rank's PCM slice == the rank's rows of the whole-file fbank (bit for bit)."""
    pcm = synth.synth_audio(23.7, seed=5)
    chunk = 500
    whole = fbank_ref.fbank(pcm)
    n_chunks = -(-whole.shape[0] // chunk)
    rows = []
    for c0, c1 in rdist.chunk_ranges(n_chunks, 3):
        s0, s1 = rdist.sample_range(len(pcm), chunk, c0, c1)
        part = fbank_ref.fbank(pcm[s0:s1])
        assert part.shape[0] == min(c1 * chunk, whole.shape[0]) - c0 * chunk
        rows.append(part)
    np.testing.assert_array_equal(np.concatenate(rows), whole)
    assert rdist.sample_range(len(pcm), chunk, n_chunks, n_chunks + 2) == (0, 0)


def _fake_results(rank):
    rng = np.random.default_rng(rank)
    out = []
    for i in range(3 + 2 * rank):
        k = int(rng.integers(0, 9))
        toks = rng.integers(1, 40, k).tolist()
        out.append(DecodeResult(tuple(toks), float(-rng.random() * 30), confidence=float(rng.random()),
                                times=sorted(rng.integers(0, 512, k).tolist()),
                                tokens_confidence=rng.random(k).tolist()))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    merged = rdist.all_gather_results(_fake_results(rank), torch.device("cpu"))
    # the gathered sequence is lazy: length, token total, random access and negative / slice indexing all agree
    assert merged.total_tokens() == sum(len(h.tokens) for h in merged)
    assert len(merged) == sum(3 + 2 * r for r in range(world))
    assert merged[-1].tokens == list(merged)[-1].tokens and [h.score for h in merged[1:4]] == [h.score for h in list(merged)[1:4]]
    q.put((rank, [(list(h.tokens), h.times, h.score, h.confidence, h.tokens_confidence) for h in merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_results_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(list(h.tokens), h.times, h.score, h.confidence, h.tokens_confidence)
            for r in range(2) for h in _fake_results(r)]
    assert got[0] == want and got[1] == want        # every rank holds all results, in chunk order


# ------------------------------------------------------------------------------------------------ diarization shard
class _StubDiarEngine:
    """Stands in for the GPU networks in the CPU test: classes and embeddings are pure functions of the window's
    samples, so a sharded run must reproduce the single-process run exactly if (and only if) every rank sees
    exactly its windows' samples."""
    frames = 589

    def __init__(self, cfg):
        self.cfg = cfg
        self.pcm = None

    def upload(self, pcm):
        self.pcm = np.asarray(pcm, np.int16)
        n, win, step = len(self.pcm), self.cfg["window_samples"], self.cfg["step_samples"]
        full = (n - win) // step + 1 if n >= win else 0
        self.n_windows = full + (1 if (n < win or (n - win) % step > 0) else 0)
        return self.n_windows

    def _window(self, w):
        win, step = self.cfg["window_samples"], self.cfg["step_samples"]
        x = np.zeros(win, np.int64)
        seg = self.pcm[w * step:w * step + win]
        x[:len(seg)] = seg
        return x

    def segment_classes(self):
        out = np.zeros((self.n_windows, self.frames), np.uint8)
        for w in range(self.n_windows):
            x = self._window(w)
            e = np.abs(x[:self.frames * 270].reshape(self.frames, 270)).mean(1)
            out[w] = np.where(e > 1500, 1 + (x[:self.frames * 270].reshape(self.frames, 270)[:, 0] > 0), 0)
        return out

    def embed(self, wins, masks):
        out = np.zeros((len(wins), self.cfg["emb_dim"]), np.float32)
        for i, (w, m) in enumerate(zip(wins, masks)):
            x = self._window(int(w)).astype(np.float64)
            base = np.zeros(self.cfg["emb_dim"])
            base[int(m[:100].sum()) % 2] = 1.0
            out[i] = base + 1e-3 * np.cos(np.arange(self.cfg["emb_dim"]) * (1 + x[::1000].sum() % 7))
        return out

    def centroid_linkage(self, X):
        from scipy.cluster.hierarchy import linkage
        return linkage(X, method="centroid", metric="euclidean")


def _diar_pipeline():
    from reverb_amd import diarization as D, synth_diar
    pipe = D.SpeakerDiarization(synth_diar.make_diar_config(), {}, {}, None)
    pipe._engine = _StubDiarEngine(pipe.cfg)
    return pipe


def _rttm(ann):
    import io
    buf = io.StringIO(); ann.write_rttm(buf)
    return buf.getvalue()


def _diar_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from reverb_amd import synth_diar
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    ann = rdist.diarize_sharded(_diar_pipeline(), pcm, torch.device("cpu"), uri="talk")
    q.put((rank, _rttm(ann)))
    dist.barrier()
    dist.destroy_process_group()


def test_window_sample_ranges():
    assert rdist.window_sample_range(1000000, 160000, 16000, 0, 3) == (0, 192000)
    assert rdist.window_sample_range(1000000, 160000, 16000, 3, 5) == (48000, 224000)
    assert rdist.window_sample_range(200000, 160000, 16000, 2, 4) == (32000, 200000)     # tail window: to the end of file
    assert rdist.window_sample_range(200000, 160000, 16000, 4, 4) == (0, 0)


def test_diarize_sharded_world2_gloo_matches_single_process():
    from reverb_amd import synth_diar
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    pipe = _diar_pipeline()
    classes, emb = pipe.networks(pcm)
    want = _rttm(pipe.finish(classes, emb, "talk"))
    assert want.count("\n") >= 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == want and got[1] == want
