#!/usr/bin/env python
"""BASELINE.json configs[3]: the diarization pipeline (pyannote SincNet+LSTM segmentation, ResNet34
speaker embeddings, clustering, RTTM turns) on long-form synthetic 16 kHz audio.  Same output contract as
bench.py; the headline metric of the repo stays bench.py's ASR RTFx -- this is the second workload.

A "step" is one full `pipeline(audio)` call on `--hours` of audio held in host memory (PCM upload
included): sinc bank + segmentation of every 10 s window (1 s hop) -> powerset classes -> masked ResNet34
embeddings of every active (window, speaker) -> GPU centroid-linkage clustering -> reconstruction.
With N > 1 (torchrun, one process per GPU) the windows are sharded in contiguous ranges and ONE RCCL
all-gather moves classes + embeddings (reverb_amd/dist.py:diarize_sharded); strong scaling (one file).

  python bench_diar.py --steps 3 --warmup 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
CONV_KEYS = ("emb_conv_32", "emb_conv_64", "emb_conv_128", "emb_conv_256", "emb_conv_s2_64", "emb_conv_s2_128",
             "emb_conv_s2_256", "emb_conv_sc")
STAGE_KEYS = ("h2d", "sinc_conv", "window_stats", "pool_norm", "sincnet_conv", "lstm_inproj", "lstm_recurrence", "linear",
              "classifier", "emb_fbank", "emb_stem") + CONV_KEYS + ("emb_pool", "emb_linear", "linkage")


def cpu_baseline(cfg, seg_sd, emb_sd, pcm, n_windows):
    """The oracle (oracle/diar_ref.py: torch restatement of PyanNet + WeSpeaker ResNet34, i.e. what pyannote runs on
    CPU) on the first windows -- the ResNet once per (window, ACTIVE local speaker) as pyannote does -- followed by the
    clustering pyannote runs on those embeddings (scipy centroid linkage + fcluster)."""
    import torch
    from scipy.cluster.hierarchy import fcluster, linkage
    from oracle import diar_ref as R
    cores = min(os.cpu_count() or 1, 32)      # more threads than that only slows the small convolutions down
    torch.set_num_threads(cores)
    win, step = cfg["window_samples"], cfg["step_samples"]
    wav = torch.from_numpy(pcm[: (n_windows - 1) * step + win].astype(np.float32) / 32768.0)
    x = torch.stack([wav[w * step:w * step + win] for w in range(n_windows)])[:, None]
    ssd, esd = R.to_torch_sd(seg_sd), R.to_torch_sd(emb_sd)
    t0 = time.perf_counter()
    embs, passes = [], 0
    with torch.no_grad():
        logp = R.pyannet(ssd, x)
        ml = torch.from_numpy(R.powerset_to_multilabel(logp))
        for w in range(n_windows):
            active = [k for k in range(ml.shape[2]) if float(ml[w, :, k].sum()) > 0]
            if not active:
                continue
            feats = torch.from_numpy(R.hamming_fbank(x[w, 0].numpy()))[None].repeat(len(active), 1, 1)
            e = R.wespeaker_embed(esd, feats, ml[w].T[active].contiguous())
            embs.append(np.asarray(e, np.float64).reshape(len(active), -1))
            passes += len(active)
    t1 = time.perf_counter()
    n_emb = 0
    if embs:
        X = np.concatenate(embs)
        X = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-12)
        n_emb = len(X)
        if n_emb > 1:
            fcluster(linkage(X, method="centroid", metric="euclidean"), 0.7045654963945799, criterion="distance")
    t2 = time.perf_counter()
    dt = t2 - t0
    return {"value": round(n_windows * (step / cfg["sample_rate"]) / dt, 3), "unit": "audio-sec/wall-sec", "cores": cores,
            "kind": "port", "sample": f"{n_windows} windows of 10 s (1 s hop = {n_windows} s of new audio): segmentation + {passes} "
                                      f"ResNet34 passes (one per active local speaker), torch fp32, {t1 - t0:.1f} s; scipy centroid "
                                      f"linkage + fcluster of the {n_emb} embeddings {1e3 * (t2 - t1):.1f} ms"}


def conv_traffic(hours, dtype):
    """HBM bytes per convolution launch (ResNet34 trunk: conv_kernel / conv_stream_kernel / conv_igemm_kernel) from two nested
    rocprofv3 --pmc passes over one step of this command (see bench.pmc_traffic)."""
    import bench
    sub = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "0", "--hours", str(hours),
           "--dtype", dtype, "--cpu-baseline-windows", "0", "--traffic", "off"]
    return bench.pmc_traffic(sub, ("conv_kernel", "conv_stream", "conv_igemm", "conv_block"), "bench_diar.py")


def run(device, rank=0, world=1, dist=None, steps=3, warmup=1, hours=1.0, dtype="bf16", cpu_windows=16, traffic="off", pipeline_steps=4):
    """Time `steps` diarization passes on `device`; returns the JSON record on rank 0 (None elsewhere).  bench.py calls
    this for its `diarization` sub-record, main() below for the stand-alone line."""
    import torch
    from reverb_amd import diarization as D, synth_diar as SD
    from reverb_amd.dist import diarize_sharded
    use_dist = dist is not None
    cfg = SD.make_diar_config()
    seg_sd, emb_sd = SD.make_segmentation_sd(cfg, 0), SD.make_embedding_sd(cfg, 0)
    pipe = D.SpeakerDiarization(cfg, seg_sd, emb_sd, None, dtype=dtype).to(device)
    base = SD.synth_conversation(120.0)
    n = int(hours * 3600 * 16000)
    pcm = np.tile(base, n // len(base) + 1)[:n]
    # de-duplicate the tiles: +-3 LSB of noise, so that no two windows (and no two embeddings) are bit-identical
    pcm = (pcm.astype(np.int32) + np.random.default_rng(7 + rank).integers(-3, 4, size=n)).clip(-32768, 32767).astype(np.int16)

    def step(resident=True):
        # one recording through the whole pipeline.  Single GPU: the samples are resident in HBM when the timed region starts (the
        # bench contract; the ASR step's convention) -- rvd_rerun_resident runs the front end (waveform, sinc filter bank, fbank)
        # on them again; `pcie_inclusive` below is the same step with the 115 MB/h upload from pageable host memory inside it.
        if use_dist:
            return diarize_sharded(pipe, pcm, device, uri="bench")
        t = time.perf_counter()
        classes, emb = pipe.networks(pcm, resident=resident)
        ann = pipe.finish(classes, emb, "bench")
        pipe.timings["total"] = time.perf_counter() - t
        return ann

    for _ in range(max(warmup, 1)):
        step(resident=False)            # (the first of these uploads the recording)
    eng = pipe.engine
    eng.reset_timings(); eng.set_profiling(True)
    comm = None
    if use_dist:       # barriers and the time reduction through librvb's communicator (the one diarize_sharded gathers on)
        from reverb_amd.dist import default_comm
        comm = default_comm(pipe.device_index if pipe.device_index is not None else device)

    def barrier():
        if comm is not None:
            comm.barrier()
        else:
            dist.barrier()
    if use_dist:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ann = step()
    torch.cuda.synchronize()
    if use_dist:
        barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        if comm is not None:
            dt = comm.max(dt)
        else:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
    eng.set_profiling(False)
    # the kernel timings and FLOP counts of the timed steps: read BEFORE the pcie_inclusive steps below (the engine keeps counting
    # FLOPs and launches with profiling off)
    conv = [eng.timing(k) for k in CONV_KEYS]
    stage_ms = {k: round(eng.timing(k)[0] / steps, 3) for k in STAGE_KEYS}
    host_s = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipe.timings.items()}
    pcie = None
    if not use_dist:
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(2):
            step(resident=False)
        torch.cuda.synchronize()
        dp = (time.perf_counter() - tp) / 2
        pcie = {"value": round(hours * 3600 / dp, 2), "ms_per_step": round(dp * 1e3, 2), "h2d_bytes_per_step": int(n) * 2,
                "host_memory": "pageable (numpy array handed to rvd_upload_pcm)", "steps": 2}
    pipelined = None
    if not use_dist and pipeline_steps > 0:
        # Throughput of a service that diarizes recording after recording: two engines alternate, and the clustering + host part of
        # recording i (a helper thread; its merge loop on 4 workgroups, as in the joint pipeline) runs UNDER the networks of
        # recording i + 1.  Reported beside `value`, which stays the sequential step.
        try:
            from concurrent.futures import ThreadPoolExecutor
            pipe2 = D.SpeakerDiarization(cfg, seg_sd, emb_sd, None, dtype=dtype).to(device)
            pipes = (pipe, pipe2)
            c2, e2 = pipe2.networks(pcm)                       # uploads the recording to the second engine; warms it
            pipe2.finish(c2, e2, "bench")
            for q in pipes:
                q.engine.set_linkage_workgroups(4)
            pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="rvd-finish")
            torch.cuda.synchronize()
            tq = time.perf_counter()
            fut = None
            for i in range(pipeline_steps):
                q = pipes[i & 1]
                ci, ei = q.networks(pcm, resident=True)
                if fut is not None:
                    fut.result()
                fut = pool.submit(q.finish, ci, ei, "bench")
            ann_p = fut.result()
            torch.cuda.synchronize()
            dq = (time.perf_counter() - tq) / pipeline_steps
            pool.shutdown()
            for q in pipes:
                q.engine.set_linkage_workgroups(0)
            pipelined = {"value": round(hours * 3600 / dq, 2), "ms_per_step": round(dq * 1e3, 2), "steps": pipeline_steps,
                         "turns": len(ann_p), "how": "two engines alternate; clustering (4 workgroups) + host part of recording i on a helper "
                                                     "thread under the networks of recording i + 1; samples resident"}
            pipe2.engine.close()
            pipe2._engine = None
        except Exception as ex:          # a sub-record must not cost the line its headline
            pipelined = {"error": f"{type(ex).__name__}: {ex}"}
    out = None
    if rank == 0:
        ms, fl, launches = sum(c[0] for c in conv), sum(c[1] for c in conv), sum(c[2] for c in conv)
        ach = fl / (ms * 1e-3) / 1e12 if ms else 0.0
        out = {
            "metric": "RTFx (audio-sec/wall-sec) diarization pipeline (pyannote segmentation + ResNet34 embeddings + clustering)",
            "value": round(hours * 3600 * steps / dt, 2), "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"diarization of one {hours:g} h 16 kHz recording: {eng.num_windows(n)} windows of 10 s / 1 s hop, "
                                   "PyanNet segmentation (SincNet + 4xBiLSTM128), WeSpeaker ResNet34 embeddings, centroid linkage, "
                                   "synthetic weights", "parallelism": f"window-shard x{world}",
                       "turns": len(ann), "speakers": len(ann.labels())},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_TFLOPS[dtype], 4), "traffic": None,
                         "kernel": "rvb::conv_kernel / conv_igemm_kernel (all ResNet34 3x3/1x1 convolutions of the timed steps, rank 0)",
                         "launches": launches, "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2),
                         "flops_per_launch": round(fl / max(launches, 1), 1)},
            "stage_ms_per_step": stage_ms,
            "host_s_last_step": host_s,
            "input": ("int16 PCM resident in HBM when the timed region starts; the step re-runs the whole front end on it"
                      if not use_dist else "every rank uploads its slice inside the step"),
            "pcie_inclusive": pcie,
            "pipelined": pipelined,
        }
        out["cpu_baseline"] = cpu_baseline(cfg, seg_sd, emb_sd, pcm, cpu_windows) if (world == 1 and cpu_windows > 0) else None
    eng.close()
    pipe._engine = None
    if out is not None and world == 1 and traffic == "auto":      # the GPU is idle now: nested PMC passes of the same workload
        t = conv_traffic(hours, dtype)
        if t is not None:
            out["roofline"]["traffic"] = t["bytes_per_launch"]
            out["roofline"]["traffic_detail"] = t
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--hours", type=float, default=1.0)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--cpu-baseline-windows", type=int, default=16)
    p.add_argument("--traffic", default="auto", choices=["auto", "off"],
                   help="auto: at N = 1 measure the convolutions' HBM traffic with two nested rocprofv3 --pmc passes")
    args = p.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # same self-launch as bench.py
        import bench
        sys.exit(bench.launch_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    use_dist = world > 1 or bool(os.environ.get("RVB_FORCE_DIST"))      # see bench.py
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if os.environ.get("RVB_COMM", "cabi") == "torch":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never depend on the container's hostname resolving
            os.environ["RVB_COMM"] = "cabi"       # tells reverb_amd.dist that this CPU group is a rendezvous: collectives are librvb's
            dist.init_process_group("gloo")       # rendezvous only (see bench.py): the collectives are librvb's
        world = dist.get_world_size()
    if not torch.cuda.is_available():
        raise SystemExit("bench_diar.py needs an MI355X: the networks have no CPU fallback")
    device = torch.device("cuda", local_rank)
    out = run(device, rank, world, dist if use_dist else None, args.steps, args.warmup, args.hours, args.dtype,
              args.cpu_baseline_windows, args.traffic)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # C-side stdout (RCCL banner) first: the JSON line is the last line
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
