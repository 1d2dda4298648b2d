#!/usr/bin/env python
"""Headline benchmark: RTFx (audio-seconds / wall-seconds) of Reverb-ASR attention_rescoring on
long-form synthetic 16 kHz audio, chunk-sharded across N MI355X (BASELINE.json configs[1]/[2]).

A "step" is one pass of the hot path over this rank's audio, PCM already resident in HBM:
  fbank -> conv subsampling -> 18 conformer blocks -> CTC head/top-k -> native prefix beam search
  -> attention rescoring -> DecodeResults on the host (+ ONE RCCL all-gather of the per-chunk
  results when N > 1).
Weak scaling: every rank decodes its own `--hours` of audio (8 h over 8 GPUs = configs[2]); chunks
are independent (reverb.py:148-180) so there is no data-path collective, only the final gather.

  python bench.py --gpus N --steps K --warmup W          # N > 1: launches N ranks itself, one per GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          # or under torchrun

At N = 1 the same JSON line also carries
  roofline.traffic  HBM bytes per GEMM launch, from two nested `rocprofv3 --pmc` passes of this very command
                    (FETCH_SIZE, WRITE_SIZE; corrected as MI355X_MICROARCH.md prescribes), `--traffic off` skips them
  pcie_inclusive    the same step with the 115 MB/h PCM upload from host memory inside it (never `value`)
  diarization       BASELINE configs[3] as a sub-record (bench_diar.py's step: own value, roofline, cpu_baseline)
  joint_fp8         BASELINE configs[4] as a sub-record (bench_joint.py: 3 h recording, ASR + diarization with their fp8 GEMMs,
                    word->speaker join; carries sharded_s / replicated_s for the 8-GPU arithmetic)
  parity_f32, asr_fp8, r268   the headline step in the exact-parity mode, with fp8 GEMMs, and on the smaller planning point
  two_streams                 the headline hour as two half-hour engines on two HIP streams (throughput of a service, no roofline)
  cpu_baseline      the oracle (CPU port of the reference) on the first chunks of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
STUB = bool(os.environ.get("RVB_BENCH_STUB"))     # CPU test hook of the launcher: gloo + a host stub instead of the engine


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="r640", choices=["tiny", "small", "r268", "r640"],
                   help="synthetic planning point (SURVEY.md section 8): r640 = d1024/16h/ff4096, 665 M params")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                   help="fp8 = the bf16 engine with the encoder's feed-forward / qkv / pointwise GEMMs in fp8 (BASELINE configs[4]); "
                        "the warm-up step calibrates the activation scales")
    p.add_argument("--hours", type=float, default=1.0, help="audio per GPU")
    p.add_argument("--chunks-per-launch", type=int, default=0, help="device batch (0 = all chunks of the audio)")
    p.add_argument("--beam", type=int, default=10)
    p.add_argument("--ctc-weight", type=float, default=0.1)
    p.add_argument("--reverse-weight", type=float, default=0.0)
    p.add_argument("--cpu-baseline-chunks", type=int, default=8, help="0 disables the CPU baseline leg (BASELINE.md section 3: >= 8 chunks)")
    p.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    p.add_argument("--traffic", default="auto", choices=["auto", "off"],
                   help="auto: at N = 1 measure the GEMMs' HBM traffic with two nested rocprofv3 --pmc passes")
    p.add_argument("--gather", default="results", choices=["results", "posteriors"],
                   help="what the ranks exchange per step: the decoded results (1.3 KB per chunk, the product path) or ALSO the "
                        "top-beam CTC posteriors (int32 id + fp32 log-prob x beam per encoder frame, 41 KB per chunk) -- the "
                        "alternative exchange of SURVEY 8(e), inside the timed region; either way a device-to-device "
                        "all-gather of that posterior payload is timed once outside it (`xgmi_allgather` in the line)")
    p.add_argument("--no-diarization", action="store_true", help="skip the diarization sub-record (N = 1 only)")
    p.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive leg (N = 1 only)")
    p.add_argument("--no-variants", action="store_true", help="skip the parity_f32 / asr_fp8 / r268 sub-records (N = 1 only)")
    p.add_argument("--joint-hours", type=float, default=3.0, help="length of the joint_fp8 sub-record's recording (BASELINE configs[4]: 3 h)")
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n: int, argv, script: str = None) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script, one per GPU, with the rendezvous
    variables torchrun would set; rank 0 inherits stdout (its JSON line stays the last line), the other ranks' stdout goes
    to stderr.  Returns the first non-zero exit code (the remaining ranks are then terminated by PID)."""
    if not STUB:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible\n")
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            c = procs[r].poll()
            if c is None:
                continue
            pending.discard(r)
            if c != 0 and rc == 0:
                rc = c
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(cfg, sd, feats_chunks, lens, args):
    """The oracle (CPU restatement of the reference, plain torch fp32, batch 1 as the reference does,
    recognize_wav.py:60-64) timed on the host cores on a bounded sample of the same workload.  The reference gets its best
    thread count: the first chunk is decoded once per candidate (8, 16, 32, 64 threads; the first decode
    of all is a discarded warm-up), the fastest candidate then decodes the whole sample."""
    import torch
    from oracle import model_ref as M, search_ref as S
    tsd = M.to_torch_sd(sd)
    cat = torch.tensor([1.0, 0.0])

    def decode(i):
        S.decode(tsd, cfg, ["attention_rescoring"], torch.from_numpy(feats_chunks[i:i + 1]),
                 torch.from_numpy(lens[i:i + 1]), args.beam, ctc_weight=args.ctc_weight,
                 reverse_weight=args.reverse_weight, cat_embs=cat)

    ncpu = os.cpu_count() or 1
    default = torch.get_num_threads()
    # candidates stop at 64 threads: on the 256-thread host of the GPU box one chunk took 1.5 s on 8 threads, 1.0 s on 32, 4.2 s on
    # 128 and 174 s (!) on 256 (profiles/r06_call3_*): torch's intra-op pool thrashes on a batch-1 model far above its sweet spot
    cands = sorted({c for c in (8, 16, 32, 64) if c <= ncpu})
    probe = {}
    if len(cands) > 1 and len(lens) > 1:
        torch.set_num_threads(cands[-1])
        decode(0)                                     # warm-up: allocator, oneDNN primitive caches
        for c in cands:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            decode(0)
            probe[c] = time.perf_counter() - t0
        cores = min(probe, key=probe.get)
    else:
        cores = cands[-1]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    frames = 0
    for i in range(len(lens)):
        decode(i)
        frames += int(lens[i])
    dt = time.perf_counter() - t0
    torch.set_num_threads(default)
    return {"value": round(frames * 0.01 / dt, 3), "unit": "RTFx (audio-sec/wall-sec)", "cores": cores, "kind": "port",
            "sample": f"first {len(lens)} chunks ({frames * 0.01:.1f} s of audio) of the same workload, oracle port, "
                      f"fp32 batch 1, {dt:.1f} s wall",
            "threads_probe_ms_chunk0": {str(c): round(1e3 * t) for c, t in probe.items()}}


# ------------------------------------------------------------------------------------------------ HBM traffic (PMC)
def pmc_traffic(sub, match, what, last=None):
    """HBM bytes per launch of the kernels whose name contains one of `match`, over one run of the command `sub`:
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (TCC slots, MI355X_MICROARCH.md); FETCH_SIZE
    doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B), both counters in units of 1024 B.
    `last` = N: only the LAST N matching dispatches of the run count (one steady-state step: a process's first step also
    launches what only a first step does).  Returns a dict or None."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    import csv
    tmp = tempfile.mkdtemp(prefix="rvb_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RVB_FORCE_DIST")}
    env["TMPDIR"] = "/tmp"
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + sub, cwd=tmp, env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            path = None
            for d, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(d, f)
            if r.returncode != 0 or path is None:
                return None
            per = {}                                   # dispatch id -> counter value (summed over the rows of one dispatch)
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if any(m in row["Kernel_Name"] for m in match) and row["Counter_Name"] == counter:
                        k = int(row["Dispatch_Id"])
                        per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            ids = sorted(per)
            if last is not None:
                ids = ids[-int(last):]
            got[counter] = (sum(per[k] for k in ids) * 1024.0, len(ids))
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (f, nf), (w, nw) = got["FETCH_SIZE"], got["WRITE_SIZE"]
    return {"bytes_per_launch": round((2.0 * f) / nf + w / nw, 1), "read_bytes_per_launch": round(2.0 * f / nf, 1),
            "write_bytes_per_launch": round(w / nw, 1), "launches": nf,
            "method": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over {what}, nested "
                      "in this run; FETCH_SIZE x2 (gfx950 wide-read correction), units of 1024 B"}


def measure_traffic(args, per_step):
    """HBM bytes per GEMM launch of this workload: this same command (headline leg only) under the two PMC passes, one warm-up
    step and one timed step; the last `per_step` GEMM dispatches -- the timed step's -- are what counts."""
    sub = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "1", "--model", args.model,
           "--dtype", args.dtype, "--hours", str(args.hours), "--chunks-per-launch", str(args.chunks_per_launch),
           "--beam", str(args.beam), "--ctc-weight", str(args.ctc_weight), "--reverse-weight", str(args.reverse_weight),
           "--cpu-baseline-chunks", "0", "--no-profile", "--traffic", "off", "--no-diarization", "--no-pcie", "--no-variants"]
    return pmc_traffic(sub, ("gemm",), "the last step of this command", last=per_step)


# ------------------------------------------------------------------------------------------------ stub (CPU test hook)
class _StubEngine:
    """Stands in for the engine when RVB_BENCH_STUB is set (tests/test_bench_contract.py runs `bench.py --gpus 2` on a
    CPU-only box): results are a pure function of (rank, chunk), there is no device work and no roofline."""

    def __init__(self, rank, n_chunks):
        self.rank, self.n_chunks = rank, n_chunks

    def decode(self):
        from reverb_amd.search import DecodeResult
        out = []
        for c in range(self.n_chunks):
            k = 3 + (c + self.rank) % 5
            out.append(DecodeResult(tuple(range(1 + self.rank, 1 + self.rank + k)), -1.0 * c, confidence=0.5,
                                    times=list(range(k)), tokens_confidence=[0.25] * k))
        time.sleep(0.01)
        return out


# ------------------------------------------------------------------------------------------------ diarization sub-record
def diarization_record(device, steps=3, warmup=2, hours=1.0, dtype="bf16", cpu_windows=16, traffic="auto"):
    """BASELINE configs[3] on this GPU, measured exactly as bench_diar.py does (one step = the whole pipeline on `hours` of
    audio whose samples are resident in HBM; `pcie_inclusive` and `pipelined` beside it); returned as a sub-record of the main
    line so that the driver's run carries it."""
    import bench_diar
    return bench_diar.run(device, rank=0, world=1, dist=None, steps=steps, warmup=warmup, hours=hours, dtype=dtype,
                          cpu_windows=cpu_windows, traffic=traffic)


# ------------------------------------------------------------------------------------------------ ASR variants (sub-records)
def asr_variant(local_rank, model, dtype, hours, beam, ctc_weight, reverse_weight, steps=2, warmup=1, state=None, note=None):
    """The headline step (fbank -> encoder -> CTC search -> rescoring -> host results, PCM resident) for another (model, dtype):
    a compact sub-record -- RTFx, ms per step and the GEMM roofline fraction against that dtype's dense MFMA peak, measured
    the same way as the main line (HIP events around the GEMM launches of the timed steps)."""
    import torch
    from reverb_amd import synth
    from reverb_amd.engine import Engine
    chunk = 2051
    seconds = hours * 3600.0
    n_samples = int(round(seconds * 16000))
    n_chunks = -(-(1 + (n_samples - 400) // 160) // chunk)
    cfg, sd = state if state is not None else synth.calibrated_state_dict(model, 0)
    eng = Engine(cfg, sd, dtype=dtype, device=local_rank, max_chunks=n_chunks, chunk_frames=chunk)
    del sd
    try:
        pcm = eng.pinned_pcm(n_samples)
        pcm[:] = synth.synth_audio(seconds, seed=1234)
        eng.upload_pcm(pcm)

        def step():
            return eng.decode_resident(eng.fbank(), ["attention_rescoring"], chunk, beam, ctc_weight, reverse_weight)["attention_rescoring"]

        for _ in range(max(warmup, 1)):          # fp8: the first pass calibrates the activation scales (and runs in bf16)
            step()
        eng.reset_timings()
        eng.set_profiling(True, gemm_only=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            hyps = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.set_profiling(False)
        g, g8 = eng.timing("gemm"), eng.timing("gemm_fp8")
    finally:
        eng.close()
    rec = {"value": round(seconds * steps / dt, 2), "unit": "audio-sec/wall-sec", "ms_per_step": round(dt / steps * 1e3, 2),
           "steps": steps, "warmup": max(warmup, 1), "dtype": dtype,
           "workload": f"Reverb-ASR attention_rescoring, {hours:g} h in {n_chunks} chunks, synthetic {model} weights "
                       f"(d={cfg['encoder_conf']['output_size']}), beam {beam}",
           "tokens_per_step": int(sum(len(h.tokens) for h in hyps))}
    k, peak = (g8, PEAK_TFLOPS["fp8"]) if dtype == "fp8" and g8 and g8["ms"] > 0 else (g, PEAK_TFLOPS["bf16" if dtype == "fp8" else dtype])
    if k and k["ms"] > 0:
        ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
        rec["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "kernel": "fp8 GEMM launches" if k is g8 else f"{'bf16' if dtype == 'fp8' else dtype} GEMM launches",
                           "gemm_ms_per_step": round(k["ms"] / steps, 3)}
        if k is g8 and g and g["ms"] > 0:
            rec["roofline"]["bf16_gemm_ms_per_step"] = round(g["ms"] / steps, 3)
    if note:
        rec["note"] = note
    return rec


def asr_two_streams(local_rank, model, dtype, hours, beam, ctc_weight, reverse_weight, steps=4, warmup=2, state=None):
    """The headline hour as TWO engines with half of the chunks each, on their own HIP streams, driven by two host threads
    (a service that decodes two recordings at a time).  The tile-count tails of the N = 1024 GEMMs (1 408 tiles on 256 CUs =
    5.5 rounds), the lock-step epilogue bursts and the host's share of the search fill up with the other stream's work:
    about 4 % more audio per second than one engine with the whole hour (profiles/r06_call10_*).  A sub-record, not the headline:
    two streams share the GPU, so a GEMM launch's duration no longer measures the kernel (no per-launch roofline here)."""
    import threading
    import torch
    from reverb_amd import synth
    from reverb_amd.engine import Engine
    chunk = 2051
    seconds = hours * 3600.0 / 2
    n_samples = int(round(seconds * 16000))
    n_chunks = -(-(1 + (n_samples - 400) // 160) // chunk)
    cfg, sd = state if state is not None else synth.calibrated_state_dict(model, 0)
    engs = [Engine(cfg, sd, dtype=dtype, device=local_rank, max_chunks=n_chunks, chunk_frames=chunk) for _ in range(2)]
    del sd
    try:
        for i, eng in enumerate(engs):
            pcm = eng.pinned_pcm(n_samples)
            pcm[:] = synth.synth_audio(seconds, seed=1234 + i)
            eng.upload_pcm(pcm)
        toks, errs = [0, 0], []

        def work(i, k):
            try:
                for _ in range(k):
                    hyps = engs[i].decode_resident(engs[i].fbank(), ["attention_rescoring"], chunk, beam, ctc_weight, reverse_weight)
                    toks[i] = sum(len(h.tokens) for h in hyps["attention_rescoring"])
            except Exception as ex:        # noqa: BLE001 -- reported by the caller's thread
                errs.append(ex)

        def run(k):
            th = [threading.Thread(target=work, args=(i, k)) for i in range(2)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            if errs:
                raise errs[0]
            return time.perf_counter() - t0

        run(max(warmup, 1))
        dt = run(steps)
    finally:
        for eng in engs:
            eng.close()
    return {"value": round(2 * seconds * steps / dt, 2), "unit": "audio-sec/wall-sec", "ms_per_step": round(dt / steps * 1e3, 2),
            "steps": steps, "warmup": max(warmup, 1), "dtype": dtype, "tokens_per_step": int(sum(toks)),
            "workload": f"the headline hour as two engines x {n_chunks} chunks, one HIP stream and one host thread each",
            "note": "throughput of two concurrent half-hour recordings; per-launch kernel durations overlap, so no roofline object"}


# ------------------------------------------------------------------------------------------------ the line
def write_long_form(out):
    """The full record (every sub-record with its stage tables, methods and notes) goes to gpurun_out/bench_long.json (or
    $RVB_BENCH_LONG) and to stderr; stdout carries the compact line below."""
    path = os.environ.get("RVB_BENCH_LONG") or os.path.join(ROOT, "gpurun_out", "bench_long.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def _brief(rec, extra=()):
    """{value, ms_per_step, frac} of a sub-record (+ the named scalars), or its error."""
    if not isinstance(rec, dict):
        return None
    if "error" in rec:
        return {"error": str(rec["error"])[:80]}
    b = {"value": rec.get("value"), "ms_per_step": rec.get("ms_per_step")}
    if isinstance(rec.get("roofline"), dict):
        b["frac"] = rec["roofline"].get("frac")
    for k in extra:
        if rec.get(k) is not None:
            b[k] = rec[k]
    return b


def compact(out, long_path=None):
    """The ONE stdout line, small enough (< 2 KB) for the driver's record to hold it whole (the 14 KB line
    lost the diarization sub-record): the contract keys, the workload, `roofline` and `cpu_baseline` in full (scalars only),
    every other sub-record as {value, ms_per_step, frac}.  configs[3] (`diarization`) and configs[4] (`joint_fp8`) close the
    line."""
    if out is None:
        return None
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "parallelism", "world_size_reported_by_process_group", "results_gathered")
                      if cfg.get(k) is not None}
    if cfg.get("backend"):
        line["config"]["backend"] = cfg["backend"][:60]
    if cfg.get("comm_fallback"):
        line["config"]["comm_fallback"] = cfg["comm_fallback"][:160]
    roof = out.get("roofline")
    line["roofline"] = {k: v for k, v in roof.items() if not isinstance(v, (dict, list)) and k != "kernel"} if roof else None
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = dict(cb, sample=cb["sample"][:130]) if cb else None
    if out.get("xgmi_allgather"):
        line["xgmi_allgather"] = {k: out["xgmi_allgather"][k] for k in ("bytes_per_rank", "ms", "busbw_GBps")}
    for k in ("pcie_inclusive", "two_streams", "parity_f32", "asr_fp8", "r268"):
        if out.get(k) is not None:
            line[k] = _brief(out[k])
    if out.get("diarization") is not None:
        line["diarization"] = _brief(out["diarization"])
    if out.get("joint_fp8") is not None:
        line["joint_fp8"] = _brief(out["joint_fp8"], ("sharded_s", "replicated_s", "projected_8gpu_step_s"))
    if long_path:
        line["long_form"] = long_path
    return line


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # RVB_FORCE_DIST=1 runs the collective code path (RCCL init, barrier, all-gather of results, max-reduce of the time)
    # even with one rank: the only way to execute it on a 1-GPU box
    use_dist = world > 1 or bool(os.environ.get("RVB_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        if STUB:
            dist.init_process_group("gloo")
        elif os.environ.get("RVB_COMM", "cabi") == "torch":      # everything through torch.distributed's own RCCL communicator
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # round 4: torch.distributed is the RENDEZVOUS only (a CPU gloo group that carries the 128-byte RCCL id once, and
            # whose store is the side channel of the failure path); every collective of the run -- the result gather, the
            # barriers around the timed region, the maximum of the step time -- goes through librvb's own communicator
            # (rvb_comm_*), so a rank holds ONE RCCL communicator
            torch.cuda.set_device(local_rank)
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never depend on the container's hostname resolving
            os.environ["RVB_COMM"] = "cabi"       # tells reverb_amd.dist that this CPU group is a rendezvous: collectives are librvb's
            dist.init_process_group("gloo")
        world = dist.get_world_size()          # what the process group reports, not what the environment claims
    if not STUB and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cpu") if STUB else torch.device("cuda", local_rank)
    if not STUB:
        torch.cuda.set_device(device)

    from reverb_amd import synth
    from reverb_amd.dist import all_gather_results, comm_fallback_reason, comm_init_hung

    chunk = 2051
    seconds = args.hours * 3600.0
    n_samples = int(round(seconds * 16000))
    n_frames = 1 + (n_samples - 400) // 160
    n_chunks = -(-n_frames // chunk)
    per_launch = args.chunks_per_launch or n_chunks
    modes = ["attention_rescoring"]
    if STUB:
        cfg, sd, eng, pcm = synth.make_config(args.model), None, _StubEngine(rank, n_chunks), None
    else:
        from reverb_amd.engine import Engine
        cfg, sd = synth.calibrated_state_dict(args.model, 0)
        state = (cfg, sd)                                   # the variants below reuse the weights (same model, other dtypes)
        eng = Engine(cfg, sd, dtype=args.dtype, device=local_rank, max_chunks=per_launch, chunk_frames=chunk)
        pcm = eng.pinned_pcm(n_samples)                   # page-locked host buffer the "reader" fills (rvb_host_alloc)
        pcm[:] = synth.synth_audio(seconds, seed=1234 + rank)
        eng.upload_pcm(pcm)                               # inputs resident in HBM before the timed region

    stats = {}

    comm = None
    if use_dist and not STUB:      # the result gather through librvb's own RCCL binding (RVB_COMM=torch: torch.distributed)
        from reverb_amd.dist import default_comm
        comm = default_comm(eng)

    def step(upload=False):
        if STUB:
            hyps = eng.decode()
        else:
            nf = eng.fbank()                 # upload=True: consumes the pending double-buffered upload (ordered behind it)
            if upload:
                eng.upload_pcm_async(pcm)    # the NEXT step's samples go up underneath this step's encoder (copy stream)
            hyps = eng.decode_resident(nf, modes, chunk, args.beam, args.ctc_weight, args.reverse_weight)["attention_rescoring"]
        ntok = sum(len(h.tokens) for h in hyps)
        if not STUB:
            stats["decoder_rows"], stats["decoder_pairs"] = eng.rescore_stats()
        if use_dist:      # ONE all-gather of the per-chunk results over RCCL/xGMI; the other ranks'
            hyps = all_gather_results(hyps, device, comm=comm)     # rows stay packed until somebody reads them (dist.GatheredResults)
            ntok = hyps.total_tokens()
            if args.gather == "posteriors" and comm is not None:
                # the top-beam posteriors of this rank's last launch, gathered DEVICE TO DEVICE (round 4: straight out of the
                # engine's HBM buffers into the communicator's, rvb_comm_allgather_topk; round 3 staged them through the host)
                nbytes, _ = comm.all_gather_topk(eng)
                stats["posterior_bytes_gathered"] = int(nbytes) * world
        return hyps, ntok

    def sync():
        if not STUB:
            torch.cuda.synchronize()

    def barrier():
        if comm is not None:
            comm.barrier()
        else:
            dist.barrier()

    def timed(k, **kw):
        if use_dist:
            barrier()
        if kw.get("upload"):
            eng.upload_pcm_async(pcm)        # steady state of a service: step 0's samples went up during the step before it
        sync()
        t0 = time.perf_counter()
        for i in range(k):
            r = step(**kw)                   # upload=True: each step uploads the NEXT step's samples -- k uploads in the region
        sync()
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
        return dt, r

    for _ in range(args.warmup):
        step()
    if not STUB:
        eng.reset_timings()
        eng.set_profiling(not args.no_profile, gemm_only=True)      # timed region: HIP events around the GEMM launches only
    dt, (hyps, ntok) = timed(args.steps)
    g, g8, att, stages, pcie = None, None, None, None, None
    if not STUB:
        eng.set_profiling(False)
        g = eng.timing("gemm")
        g8 = eng.timing("gemm_fp8")
        att = eng.timing("attention")
        if not args.no_profile:       # the per-stage table comes from ONE extra, untimed step with every stage bracketed
            eng.reset_timings()
            eng.set_profiling(True)
            step()
            eng.set_profiling(False)
            stages = {k: eng.timing(k) for k in ("fbank", "subsample", "gemm", "gemm_fp8", "attention", "rownorm", "glu_dwconv",
                                                 "ctc_topk", "embed", "lse_gather", "search_host")}
        if world == 1 and not args.no_pcie:
            # the same steps from PCM in (page-locked) HOST memory: SURVEY.md 8d's end-to-end definition.  Every step's
            # samples cross PCIe inside the timed region (K uploads for K steps); since round 4 they are double-buffered
            # (rvb_upload_pcm_async): step i+1's samples travel on a copy stream underneath step i's encoder -- the steady state
            # of a service that decodes recording after recording (the upload of step 0 happened during "the step before" and
            # the last step uploads the samples of a step that is not timed: K uploads for K steps inside the region).
            # Reported beside `value` (inputs HBM-resident there, as the bench contract asks).
            dp, _ = timed(args.steps, upload=True)
            pcie = {"value": round(seconds * args.steps / dp, 2), "ms_per_step": round(dp / args.steps * 1e3, 2),
                    "h2d_bytes_per_step": int(n_samples * 2), "host_memory": "page-locked (rvb_host_alloc)",
                    "uploads_in_timed_region": args.steps,
                    "overlap": "double-buffered: upload of step i+1 on a copy stream under step i (rvb_upload_pcm_async)"}

    xgmi = None
    if comm is not None:
        # SURVEY 8(e)'s bandwidth datapoint, outside the timed region: the posterior exchange of one hour per rank as ONE
        # all-gather between device buffers (every rank calls it; rank 0 reports)
        enc_frames = ((chunk - 3) // 2 - 3) // 2 + 1 if chunk >= 7 else 0
        payload = n_chunks * enc_frames * args.beam * 8
        ms = comm.time_all_gather(payload, iters=10)
        xgmi = {"what": "all-gather of the top-beam CTC posteriors of one hour per rank (int32 id + fp32 log-prob, "
                        f"{enc_frames} frames x beam {args.beam} = {enc_frames * args.beam * 8 / 1e3:.1f} KB per chunk), device to device, "
                        "HIP events, 10 calls after one warm-up; not part of `value`",
                "bytes_per_rank": payload, "ms": round(ms, 4),
                "algbw_GBps": round(payload * world / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                "busbw_GBps": round(payload * (world - 1) / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}

    out = None
    if rank == 0:
        audio_total = seconds * world * args.steps
        roof = None
        if args.dtype == "fp8" and g8 and g8["ms"] > 0:
            # the dominant kernel of this mode is the fp8 GEMM (priced against the 5 PFLOP/s dense fp8 peak); the GEMMs
            # that stay in bf16 (subsampling, attention output, CTC head, decoder) are listed beside it
            ach = g8["flops"] / (g8["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS["fp8"], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS["fp8"], 4), "traffic": None,
                    "kernel": "rvb::gemm2_kernel<fp8> (fp8 GEMM launches of the timed steps)",
                    "launches": g8["launches"], "avg_launch_us": round(g8["ms"] * 1e3 / max(g8["launches"], 1), 2),
                    "flops_per_launch": round(g8["flops"] / max(g8["launches"], 1), 1),
                    "bf16_gemms": {"achieved": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 2) if g["ms"] > 0 else None,
                                   "launches": g["launches"], "ms_per_step": round(g["ms"] / args.steps, 3)},
                    "fp8_ms_per_step": round(g8["ms"] / args.steps, 3)}
        elif g and g["ms"] > 0:
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                    "kernel": "rvb::gemm_kernel (all GEMM launches of the timed steps)",
                    "launches": g["launches"], "avg_launch_us": round(g["ms"] * 1e3 / max(g["launches"], 1), 2),
                    "flops_per_launch": round(g["flops"] / max(g["launches"], 1), 1),
                    # every operand of a launch once (A, W, bias, C, fp32 residual; rvb_get_timing_bytes): what `traffic` is read against
                    "algorithmic_bytes_per_launch": round(g["bytes"] / max(g["launches"], 1), 1)}
        if roof is not None and att is not None:
            # the whole step against the same peak: MFMA FLOPs executed (GEMMs + encoder attention) / wall time of the step
            roof["step_frac"] = round((g["flops"] + (g8["flops"] if g8 else 0.0) + att["flops"]) / dt / 1e12 / PEAK_TFLOPS["bf16" if args.dtype == "fp8" else args.dtype], 4)
        out = {
            "metric": "RTFx (audio-sec/wall-sec) Reverb-ASR attention_rescoring",
            "value": round(audio_total / dt, 2),
            "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "stub (launcher test, no device work)" if STUB else "synthetic",
            "config": {"workload": f"Reverb-ASR attention_rescoring, {args.hours:g} h of 16 kHz audio per GPU in "
                                   f"{n_chunks} chunks of 20.51 s, synthetic {args.model} weights "
                                   f"(d={cfg['encoder_conf']['output_size']}, {cfg['encoder_conf']['num_blocks']} conformer + "
                                   f"3+3 decoder blocks, vocab {cfg['output_dim']}), beam {args.beam}, "
                                   f"ctc_weight {args.ctc_weight}",
                       "reverse_weight": args.reverse_weight,
                       "chunks_per_launch": per_launch, "parallelism": f"chunk-shard x{world}",
                       "world_size_reported_by_process_group": world if use_dist else 1,
                       "backend": (("librvb rvb_comm_* (RCCL): gather, barriers, time reduction; torch.distributed " + dist.get_backend() +
                                    " = rendezvous only" if comm else dist.get_backend()) if use_dist else None),
                       "comm_fallback": comm_fallback_reason() if use_dist and not STUB else None,      # why librvb's RCCL binding is not in use
                       "gather": args.gather if use_dist else None,
                       "posterior_bytes_gathered_per_step": stats.get("posterior_bytes_gathered"),
                       "results_gathered": len(hyps), "tokens_per_step": int(ntok),
                       # rescoring: (hypothesis, position) log-probs served vs decoder rows computed (one per distinct prefix)
                       "decoder_pairs_per_step": stats.get("decoder_pairs"), "decoder_rows_per_step": stats.get("decoder_rows")},
            "roofline": roof,
            "stage_ms_per_step": {k: round(v["ms"], 3) for k, v in stages.items()} if stages else None,
            "pcie_inclusive": pcie,
            "xgmi_allgather": xgmi,
        }
        if not STUB and world == 1 and args.cpu_baseline_chunks > 0:
            nb = min(args.cpu_baseline_chunks, n_chunks)
            _, feats = eng.fbank(return_feats=True)
            x = np.zeros((nb, chunk, 80), np.float32)
            lens = np.zeros(nb, np.int32)
            for i in range(nb):
                part = feats[i * chunk:(i + 1) * chunk]
                x[i, :len(part)] = part
                lens[i] = len(part)
            out["cpu_baseline"] = cpu_baseline(cfg, sd, x, lens, args)
        else:
            out["cpu_baseline"] = None
    if comm is not None:
        comm.close()
    if not STUB:
        eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not STUB and world == 1:
        # the engine is closed and the GPU idle: nested measurements of the same workload
        if args.traffic == "auto" and out["roofline"] is not None and args.dtype != "fp8":
            per_step = out["roofline"]["launches"] // max(args.steps, 1)
            t = measure_traffic(args, per_step)
            if t is not None and t["launches"] != per_step:
                # the nested passes must have profiled exactly one step of THIS workload's GEMMs (a pass that
                # also ran the variants averaged 3 731 launches instead of 333); anything else is not this kernel's traffic
                out["roofline"]["traffic_error"] = f"nested PMC pass saw {t['launches']} GEMM launches, one step has {per_step}: dropped"
                t = None
            if t is not None:
                out["roofline"]["traffic"] = t["bytes_per_launch"]
                out["roofline"]["traffic_launches"] = t["launches"]
                out["roofline"]["traffic_over_algorithmic"] = round(t["bytes_per_launch"] / max(out["roofline"]["algorithmic_bytes_per_launch"], 1.0), 3)
                out["roofline"]["traffic_detail"] = t
        if not args.no_diarization:
            try:
                out["diarization"] = diarization_record(device, traffic=args.traffic)
            except Exception as ex:        # the headline line must not die with the second workload
                out["diarization"] = {"error": f"{type(ex).__name__}: {ex}"}
            try:                           # BASELINE configs[4]: joint pipeline on a 3 h recording, fp8 MFMA GEMMs on both sides
                import bench_joint
                out["joint_fp8"] = bench_joint.run(local_rank, steps=2, warmup=1, hours=args.joint_hours, dtype="fp8", model=args.model,
                                                   state=state)
            except Exception as ex:
                out["joint_fp8"] = {"error": f"{type(ex).__name__}: {ex}"}
        if not args.no_variants:
            # the same step in the other modes: f32 = the only mode whose token ids are bit-exact against the
            # reference (tests/test_longform_gpu.py decodes this very hour with 0 edits); fp8 = BASELINE configs[4]'s GEMM dtype
            # on the ASR side alone; r268 = the smaller planning point of SURVEY section 8 (d = 640)
            kw = dict(hours=args.hours, beam=args.beam, ctc_weight=args.ctc_weight, reverse_weight=args.reverse_weight)
            for key, model, dtype, steps, note in (
                    ("parity_f32", args.model, "f32", 1, "exact-parity mode: v_mfma_f32_16x16x4_f32, ids identical to the reference on this hour"),
                    ("asr_fp8", args.model, "fp8", 3, "default fp8 policy: feed-forward GEMMs on e4m3 operands"),
                    ("r268", "r268", args.dtype, 3, None)):
                if key == "asr_fp8" and args.dtype == "fp8" or key == "r268" and args.model == "r268":
                    continue
                try:
                    out[key] = asr_variant(local_rank, model, dtype, steps=steps, warmup=1, note=note,
                                           state=state if model == args.model else None, **kw)
                except Exception as ex:
                    out[key] = {"error": f"{type(ex).__name__}: {ex}"}
            try:
                out["two_streams"] = asr_two_streams(local_rank, args.model, args.dtype, state=state, **kw)
            except Exception as ex:
                out["two_streams"] = {"error": f"{type(ex).__name__}: {ex}"}
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in the C stdout buffer: get it out first,
        long_path = write_long_form(out)
        sys.stderr.write("bench.py long form: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        print(json.dumps(compact(out, long_path)), flush=True)  # so that the (compact) JSON line is the LAST line of stdout
    if comm_init_hung():
        os._exit(0)       # a helper thread is still inside ncclCommInitRank (the run fell back to torch.distributed): do not wait for it


if __name__ == "__main__":
    main()
