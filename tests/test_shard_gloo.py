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
        if i % 3 == 2:      # greedy-style result: no times, no confidences, the ctc_frames extension instead (engine.greedy)
            r = DecodeResult(toks)
            r.ctc_frames = sorted(rng.integers(0, 512, k).tolist())
        else:
            r = DecodeResult(tuple(toks), float(-rng.random() * 30), confidence=float(rng.random()),
                             times=sorted(rng.integers(0, 512, k).tolist()), tokens_confidence=rng.random(k).tolist())
        out.append(r)
    return out


def _row(h):
    return (list(h.tokens), h.times, h.ctc_frames, h.score, h.confidence, h.tokens_confidence)


def test_pack_unpack_roundtrip_keeps_none_and_empty_apart():
    hyps = _fake_results(0) + _fake_results(3) + [DecodeResult((), 0.0, times=[], tokens_confidence=[])]
    back = rdist.unpack_results(rdist.pack_results(hyps))
    assert [_row(h) for h in back] == [_row(h) for h in hyps]
    assert back[-1].times == [] and back[2].times is None and back[2].ctc_frames is not None


def _worker(rank, world, port, q, row_words):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if row_words:
        rdist._ROW_WORDS = row_words           # force the overflow path: the rows do not fit the first buffer
    kmax = max(3 + 2 * r for r in range(world))
    merged = rdist.all_gather_results(_fake_results(rank), torch.device("cpu"), max_count=kmax)
    first = rdist.N_COLLECTIVES
    again = rdist.all_gather_results(_fake_results(rank), torch.device("cpu"), max_count=kmax)
    # the gathered sequence is lazy: length, token total, random access and negative / slice indexing all agree
    assert merged.total_tokens() == sum(len(h.tokens) for h in merged)
    assert len(merged) == sum(3 + 2 * r for r in range(world))
    assert merged[-1].tokens == list(merged)[-1].tokens and [h.score for h in merged[1:4]] == [h.score for h in list(merged)[1:4]]
    assert [_row(h) for h in again] == [_row(h) for h in merged]
    q.put((rank, [_row(h) for h in merged], first, rdist.N_COLLECTIVES - first))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(target, args=()):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + tuple(args)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return {g[0]: g[1:] for g in got}


@pytest.mark.parametrize("row_words", [0, 6])
def test_all_gather_results_world2_gloo(row_words):
    got = _run_world2(_worker, (row_words,))
    want = [_row(h) for r in range(2) for h in _fake_results(r)]
    for r in range(2):
        rows, first, second = got[r]
        assert rows == want                          # every rank holds all results, in chunk order
        # ONE collective per call; a buffer that turns out too small costs exactly one more, once (the capacity is sticky)
        assert first == (2 if row_words else 1) and second == 1


class _FakeComm:
    """RvbComm stand-in for the fallback test: creation / the preflight collective fail on the ranks the test names."""
    fail_create, fail_preflight, closed = (), (), 0

    def __init__(self, engine, world, rank, unique_id):
        assert len(unique_id) == 128
        if rank in _FakeComm.fail_create:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")
        self.world, self.rank, self.handle = world, rank, 1

    @staticmethod
    def unique_id():
        return bytes(128)

    def set_timeout(self, s):
        pass

    def all_gather(self, send):
        if self.rank in _FakeComm.fail_preflight:
            raise RuntimeError("rvb_comm_allgather: no completion within 30 s (a peer is missing); communicator aborted")
        return np.stack([np.full_like(send, r) for r in range(self.world)])

    def close(self):
        _FakeComm.closed += 1
        self.handle = None


def _fallback_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), RVB_COMM="cabi")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.is_available = lambda: True            # "a GPU box": default_comm goes for librvb's communicator
    rdist.RvbComm = _FakeComm
    _FakeComm.fail_create = (1,) if mode == "create" else ()
    _FakeComm.fail_preflight = (0,) if mode == "preflight" else ()
    comm = rdist.default_comm(0)
    again = rdist.default_comm(0)                     # a later caller (the other engine) gets the same answer, no new attempt
    merged = rdist.all_gather_results(_fake_results(rank), torch.device("cpu"), max_count=max(3 + 2 * r for r in range(world)), comm=comm) \
        if comm is None else None
    q.put((rank, comm is None, again is None, rdist.comm_fallback_reason(), _FakeComm.closed, [_row(h) for h in merged] if merged else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["create", "preflight", "fine"])
def test_failed_rccl_init_on_one_rank_makes_every_rank_fall_back_together(mode):
    """`rvb_comm_create` with world > 1 first runs on the driver's scaling box.  A failure on ANY
    rank (creation, or the 1-KB preflight all-gather) must not raise and must not leave the ranks on different transports: all of
    them free what they created, record the reason and gather through torch.distributed; with no failure all keep librvb's."""
    got = _run_world2(_fallback_worker, (mode,))
    want = [_row(h) for r in range(2) for h in _fake_results(r)]
    for r in range(2):
        is_none, again_none, reason, closed, rows = got[r]
        if mode == "fine":
            assert not is_none and not again_none and reason is None and closed == 0
            continue
        assert is_none and again_none and rows == want
        assert ("rank 1: RuntimeError: ncclCommInitRank failed" in reason) if mode == "create" else ("rank 0: preflight all-gather failed" in reason)
    # the rank whose own communicator was fine freed it
    assert got[0 if mode == "create" else 1][3] == (1 if mode != "fine" else 0)


# ------------------------------------------------------------------------------------------------ decode_sharded
class _StubAsrEngine:
    """Stands in for the GPU engine: the result of a chunk is a pure function of the samples its frames cover, so the
    sharded run reproduces the single-process run exactly if (and only if) every rank decodes exactly its chunks from
    exactly its samples (dist.sample_range with the 240-sample halo)."""

    def upload_pcm(self, pcm):
        self.pcm = np.asarray(pcm, np.int16)

    def fbank(self):
        return rdist.num_frames(len(self.pcm))

    def decode_resident(self, n_frames, modes, chunk_size, beam_size, ctc_weight, reverse_weight, blank_penalty=0.0):
        out = {m: [] for m in modes}
        for c in range(-(-n_frames // chunk_size)):
            f0, f1 = c * chunk_size, min((c + 1) * chunk_size, n_frames)
            x = self.pcm[f0 * 160:(f1 - 1) * 160 + 400].astype(np.int64)
            k = 2 + int(np.abs(x).sum() % 7)
            toks = [int(abs(int(x[(j * 7919) % len(x)])) % 97) + 1 for j in range(k)]
            for m in modes:
                if m == "ctc_greedy_search":
                    r = DecodeResult(toks)
                    r.ctc_frames = list(range(k))
                else:
                    r = DecodeResult(tuple(toks), -float(k) - ctc_weight, confidence=1.0 / k, times=[3 * j for j in range(k)],
                                     tokens_confidence=[float(t) / 100 for t in toks])
                out[m].append(r)
        return out


_MODES = ["ctc_greedy_search", "attention_rescoring"]


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = synth.synth_audio(37.3, seed=11)
    res = rdist.decode_sharded(_StubAsrEngine(), pcm, _MODES, 500, 10, 0.1, 0.0, torch.device("cpu"))
    q.put((rank, {m: [_row(h) for h in res[m]] for m in _MODES}))
    dist.barrier()
    dist.destroy_process_group()


def test_decode_sharded_world2_gloo_matches_single_process():
    pcm = synth.synth_audio(37.3, seed=11)
    eng = _StubAsrEngine()
    eng.upload_pcm(pcm)
    want = eng.decode_resident(eng.fbank(), _MODES, 500, 10, 0.1, 0.0)
    assert len(want[_MODES[0]]) == 8                 # 3729 frames -> 7 full chunks + a tail: ranks get 4 + 4
    want = {m: [_row(h) for h in want[m]] for m in _MODES}
    got = _run_world2(_sharded_worker)
    assert got[0][0] == want and got[1][0] == want


def _dying_worker(rank, world, port, q, mode):
    """mode "kill": rank 1 joins the process group and then dies before decoding anything (os._exit: no clean-up, no goodbye);
    mode "no-collective": both ranks live, but the collective itself is made to fail on both (what an RCCL timeout looks like)."""
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=8))
    if mode == "kill" and rank == 1:
        os._exit(0)
    if mode == "no-collective":
        def broken(*a, **k):
            raise RuntimeError("collective timed out (test)")
        rdist.all_gather_results = broken
    pcm = synth.synth_audio(37.3, seed=11)
    res = rdist.decode_sharded(_StubAsrEngine(), pcm, _MODES, 500, 10, 0.1, 0.0, torch.device("cpu"), timeout=2.0)
    first = rdist.decode_sharded.last_recovery
    # the NEXT recording of the same process.  With a rank on record as dead the survivors share it among
    # themselves and exchange through the store at once -- no wait for the dead rank, nothing re-decoded
    import time
    t0 = time.time()
    res2 = rdist.decode_sharded(_StubAsrEngine(), pcm, _MODES, 500, 10, 0.1, 0.0, torch.device("cpu"), timeout=2.0)
    second = dict(rdist.decode_sharded.last_recovery or {}, seconds=time.time() - t0, known_dead=sorted(rdist._KNOWN_DEAD),
                  same={m: [_row(h) for h in res2[m]] == [_row(h) for h in res[m]] for m in _MODES})
    q.put((rank, {m: [_row(h) for h in res[m]] for m in _MODES}, dict(first, second=second)))
    if mode != "kill":
        dist.barrier()
        dist.destroy_process_group()
    else:
        q.close(); q.join_thread()          # flush the result before leaving without clean-up
        os._exit(0)                          # (the process group's destructor would wait for the dead peer)


@pytest.mark.parametrize("mode", ["kill", "no-collective"])
def test_decode_sharded_survives_a_dead_rank_and_a_dead_collective(mode):
    """SURVEY.md section 5 / VERDICT r3 "next" #6: a rank that dies must cost its chunk range, not the job.  World 2 over gloo:
    rank 1 is killed right after the rendezvous -- rank 0's gather fails (or times out), it finds rank 1's results missing
    from the store, re-decodes rank 1's chunks itself and returns the complete recording, identical to a single-process
    run.  Second case: nobody dies but the collective cannot complete on either rank -- both exchange through the store and
    nothing is re-decoded."""
    pcm = synth.synth_audio(37.3, seed=11)
    eng = _StubAsrEngine()
    eng.upload_pcm(pcm)
    want = eng.decode_resident(eng.fbank(), _MODES, 500, 10, 0.1, 0.0)
    want = {m: [_row(h) for h in want[m]] for m in _MODES}
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    n_results = 1 if mode == "kill" else 2
    got = {}
    for _ in range(n_results):
        g = q.get(timeout=180)
        got[g[0]] = g[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, (rows, info) in got.items():
        assert rows == want, f"rank {r}"
        assert info is not None and info["alive"] == ([0] if mode == "kill" else [0, 1])
        second = info["second"]
        assert all(second["same"].values())
        if mode == "kill":
            assert info["dead"] == [1] and info["plan"] == [[1, 0, 4, 8]]         # rank 1's chunks 4..7, re-queued on rank 0
            assert second["known_dead"] == [1] and second["alive"] == [0] and second["plan"] == [] and second["seconds"] < 1.5
        else:
            assert info["dead"] == [] and info["plan"] == []
            assert second["known_dead"] == []                                     # nobody died: the next call tries the collective again


def test_greedy_results_survive_the_gather_for_get_output():
    """a gathered greedy result must still carry ctc_frames (get_output falls back to them when times is None)."""
    r = DecodeResult([5, 6, 7]); r.ctc_frames = [1, 4, 9]
    back = rdist.unpack_results(rdist.pack_results([r]))[0]
    assert back.times is None and back.ctc_frames == [1, 4, 9] and list(back.tokens) == [5, 6, 7]


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


def test_diar_shard_packing_roundtrip():
    """The one buffer a rank contributes to the diarization gather: classes bytes + embedding bit patterns (NaN = inactive
    speaker survives), fixed size for (kmax, frames, dim), empty ranks included."""
    assert rdist.segmentation_frames(160000) == 589 and rdist.segmentation_frames(80000) == 293
    rng = np.random.default_rng(1)
    classes = rng.integers(0, 7, size=(4, 589)).astype(np.uint8)
    emb = rng.standard_normal((4, 3, 16)).astype(np.float32)
    emb[2, 1] = np.nan
    w = rdist.pack_diar_shard(classes, emb, 6, 589, 16)
    assert w.dtype == np.int32 and w.size == rdist.pack_diar_shard(None, None, 6, 589, 16).size
    c2, e2 = rdist.unpack_diar_shard(w, 6)
    np.testing.assert_array_equal(c2, classes)
    assert np.array_equal(np.isnan(e2), np.isnan(emb)) and np.array_equal(np.nan_to_num(e2), np.nan_to_num(emb))
    c0, e0 = rdist.unpack_diar_shard(rdist.pack_diar_shard(None, None, 6, 589, 16), 6)
    assert c0.shape == (0, 589) and e0.shape == (0, 3, 16)


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


class _StubFp8DiarEngine(_StubDiarEngine):
    """fp8 trunk stand-in: the first embed call "calibrates" scales from this rank's own samples; installed scales win."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.scales, self.embed_calls, self.installed = None, 0, 0

    def embed(self, wins, masks):
        self.embed_calls += 1
        if self.scales is None:
            self.scales = np.full(32, 2.0 ** (int(np.abs(self.pcm).max()) % 5), np.float32)
        return super().embed(wins, masks) + 1e-6 * self.scales[0]

    def emb_fp8(self):
        return (2 if self.scales is not None else 0), (self.scales if self.scales is not None else np.zeros(32, np.float32)), 0

    def set_emb_fp8_scales(self, sc):
        self.scales = np.asarray(sc, np.float32).copy()
        self.installed += 1


def _diar_fp8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from reverb_amd import synth_diar
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    pcm[: len(pcm) // 2] = (pcm[: len(pcm) // 2] // 2)          # the two halves calibrate to different ranges
    pipe = _diar_pipeline()
    pipe.dtype = "fp8"
    pipe._engine = _StubFp8DiarEngine(pipe.cfg)
    a1 = rdist.diarize_sharded(pipe, pcm, torch.device("cpu"), uri="talk")
    calls1 = pipe._engine.embed_calls
    a2 = rdist.diarize_sharded(pipe, pcm, torch.device("cpu"), uri="talk")
    q.put((rank, pipe._emb_fp8_scales.tolist(), calls1, pipe._engine.embed_calls, pipe._engine.installed, _rttm(a1), _rttm(a2)))
    dist.barrier()
    dist.destroy_process_group()


def test_diarize_sharded_fp8_ranks_agree_on_one_set_of_trunk_scales():
    """each rank of a sharded fp8 diarization calibrated its embedding scales on its own windows.  Now the first
    sharded recording of a pipeline gathers the scales, installs the element-wise maximum everywhere and embeds again; the
    second recording does neither."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diar_fp8_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sc0, c0, t0, i0, r0a, r0b), (_, sc1, c1, t1, i1, r1a, r1b) = got
    assert sc0 == sc1 and len(sc0) == 32 and i0 == i1 == 1
    assert c0 == c1 == 2 and t0 == t1 == 3                   # calibrate + re-embed on the first recording, one pass on the second
    assert r0a == r1a == r0b == r1b


def test_networks_host_overlap_keeps_results():
    """Round 4: networks() computes the pooling masks of everything behind the first trunk pass, and finish()'s speaker count /
    activity table, on a helper thread underneath the embedding call.  Whatever the split point (here: after 1 window, after 2,
    none), classes, embeddings and the RTTM are those of the unsplit run -- and finish() on a DIFFERENT classes object (what the
    sharded path hands it after the gather) falls back to computing its inputs itself."""
    from reverb_amd import synth_diar
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    outs = []
    for head in (10 ** 9, 1, 2):
        pipe = _diar_pipeline()
        pipe.HEAD_WINDOWS = head
        classes, emb = pipe.networks(pcm)
        assert pipe.timings["embeddings"] == int(np.isfinite(emb[:, :, 0]).sum())
        outs.append((classes.copy(), emb.copy(), _rttm(pipe.finish(classes, emb, "talk")), _rttm(pipe.finish(classes.copy(), emb, "talk"))))
    for c, e, r1, r2 in outs[1:]:
        np.testing.assert_array_equal(c, outs[0][0])
        assert np.array_equal(np.isnan(e), np.isnan(outs[0][1])) and np.array_equal(np.nan_to_num(e), np.nan_to_num(outs[0][1]))
        assert r1 == outs[0][2] and r2 == outs[0][2]
    assert outs[0][2] == outs[0][3] and outs[0][2].count("\n") >= 2


def _dying_diar_worker(rank, world, port, q):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from reverb_amd import synth_diar
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=8))
    if rank == 1:
        os._exit(0)                          # dies right after the rendezvous: its windows are never computed
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    ann = rdist.diarize_sharded(_diar_pipeline(), pcm, torch.device("cpu"), uri="talk", timeout=2.0)
    q.put((rank, _rttm(ann), rdist.diarize_sharded.last_recovery))
    q.close(); q.join_thread()
    os._exit(0)


def test_diarize_sharded_survives_a_dead_rank():
    """Round 4: the diarization shard has decode_sharded's failure path (dist._recover_through_store): rank 1 dies after the
    rendezvous, rank 0's gather fails, rank 0 runs rank 1's windows itself and returns the RTTM of a single-process run."""
    from reverb_amd import synth_diar
    pcm = synth_diar.synth_conversation(33.4, seed=9)
    pipe = _diar_pipeline()
    classes, emb = pipe.networks(pcm)
    want = _rttm(pipe.finish(classes, emb, "talk"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_diar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rank, got, info = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rank == 0 and got == want
    n_windows = classes.shape[0]
    assert info["alive"] == [0] and info["dead"] == [1] and info["plan"] == [[1, 0, (n_windows + 1) // 2, n_windows]]
