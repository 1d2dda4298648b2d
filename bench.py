#!/usr/bin/env python
"""Headline benchmark: RTFx (audio-seconds / wall-seconds) of Reverb-ASR attention_rescoring on
long-form synthetic 16 kHz audio, chunk-sharded across N MI355X (BASELINE.json configs[1]/[2]).

A "step" is one pass of the hot path over this rank's audio, PCM already resident in HBM:
  fbank -> conv subsampling -> 18 conformer blocks -> CTC head/top-k -> native prefix beam search
  -> attention rescoring -> DecodeResults on the host (+ one RCCL all-gather of the per-chunk
  results when N > 1).
Weak scaling: every rank decodes its own `--hours` of audio (8 h over 8 GPUs = configs[2]); chunks
are independent (reverb.py:148-180) so there is no data-path collective, only the final gather.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="r640", choices=["tiny", "small", "r268", "r640"],
                   help="synthetic planning point (SURVEY.md section 8): r640 = d1024/16h/ff4096, 665 M params")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--hours", type=float, default=1.0, help="audio per GPU")
    p.add_argument("--chunks-per-launch", type=int, default=0, help="device batch (0 = all chunks of the audio)")
    p.add_argument("--beam", type=int, default=10)
    p.add_argument("--ctc-weight", type=float, default=0.1)
    p.add_argument("--reverse-weight", type=float, default=0.0)
    p.add_argument("--cpu-baseline-chunks", type=int, default=4, help="0 disables the CPU baseline leg")
    p.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    return p.parse_args()


def cpu_baseline(cfg, sd, feats_chunks, lens, args):
    """The oracle (CPU restatement of the reference, plain torch fp32, batch 1 as the reference does,
    recognize_wav.py:60-64) timed on the host cores on a bounded sample of the same workload."""
    import torch
    from oracle import model_ref as M, search_ref as S
    tsd = M.to_torch_sd(sd)
    cores = torch.get_num_threads()
    cat = torch.tensor([1.0, 0.0])
    t0 = time.perf_counter()
    frames = 0
    for i in range(len(lens)):
        S.decode(tsd, cfg, ["attention_rescoring"], torch.from_numpy(feats_chunks[i:i + 1]),
                 torch.from_numpy(lens[i:i + 1]), args.beam, ctc_weight=args.ctc_weight,
                 reverse_weight=args.reverse_weight, cat_embs=cat)
        frames += int(lens[i])
    dt = time.perf_counter() - t0
    return {"value": round(frames * 0.01 / dt, 3), "unit": "RTFx (audio-sec/wall-sec)", "cores": cores, "kind": "port",
            "sample": f"first {len(lens)} chunks ({frames * 0.01:.1f} s of audio) of the same workload, oracle "
                      f"attention_rescoring fp32 batch 1, {dt:.1f} s wall"}


def main():
    args = parse()
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
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from reverb_amd import synth
    from reverb_amd.dist import all_gather_results
    from reverb_amd.engine import Engine

    chunk = 2051
    seconds = args.hours * 3600.0
    n_samples = int(round(seconds * 16000))
    n_frames = 1 + (n_samples - 400) // 160
    n_chunks = -(-n_frames // chunk)
    per_launch = args.chunks_per_launch or n_chunks
    cfg, sd = synth.calibrated_state_dict(args.model, 0)
    eng = Engine(cfg, sd, dtype=args.dtype, device=local_rank, max_chunks=per_launch, chunk_frames=chunk)
    pcm = synth.synth_audio(seconds, seed=1234 + rank)
    eng.upload_pcm(pcm)                                   # inputs resident in HBM before the timed region
    modes = ["attention_rescoring"]

    def step():
        nf = eng.fbank()
        hyps = eng.decode_resident(nf, modes, chunk, args.beam, args.ctc_weight, args.reverse_weight)["attention_rescoring"]
        ntok = sum(len(h.tokens) for h in hyps)
        if use_dist:      # one all-gather of the per-chunk results over RCCL/xGMI; the other ranks'
            hyps = all_gather_results(hyps, device)     # rows stay packed until somebody reads them (dist.GatheredResults)
            ntok = hyps.total_tokens()
        return hyps, ntok

    for _ in range(args.warmup):
        step()
    eng.reset_timings()
    eng.set_profiling(not args.no_profile, gemm_only=True)      # timed region: HIP events around the GEMM launches only
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hyps, ntok = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.set_profiling(False)
    g = eng.timing("gemm")
    stages = None
    if not args.no_profile:       # the per-stage table comes from ONE extra, untimed step with every stage bracketed
        eng.reset_timings()
        eng.set_profiling(True)
        step()
        eng.set_profiling(False)
        stages = {k: eng.timing(k) for k in ("fbank", "subsample", "gemm", "attention", "rownorm", "glu_dwconv",
                                             "ctc_topk", "embed", "lse_gather", "search_host")}

    if rank == 0:
        audio_total = seconds * world * args.steps
        roof = None
        traffic = None      # HBM bytes per GEMM launch from a PMC pass of this command (scripts/pmc_traffic.py)
        tpath = os.path.join(ROOT, "profiles", f"gemm_traffic_{args.model}_{args.hours:g}h_{args.dtype}.json")
        if os.path.exists(tpath):
            with open(tpath) as tf:
                traffic = round(json.load(tf)["traffic_bytes_per_launch"], 1)
        if g["ms"] > 0:
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                    "kernel": "rvb::gemm_kernel (all GEMM launches of the timed steps)",
                    "launches": g["launches"], "avg_launch_us": round(g["ms"] * 1e3 / max(g["launches"], 1), 2),
                    "flops_per_launch": round(g["flops"] / max(g["launches"], 1), 1)}
        out = {
            "metric": "RTFx (audio-sec/wall-sec) Reverb-ASR attention_rescoring",
            "value": round(audio_total / dt, 2),
            "unit": "audio-sec/wall-sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Reverb-ASR attention_rescoring, {args.hours:g} h of 16 kHz audio per GPU in "
                                   f"{n_chunks} chunks of 20.51 s, synthetic {args.model} weights "
                                   f"(d={cfg['encoder_conf']['output_size']}, {cfg['encoder_conf']['num_blocks']} conformer blocks, "
                                   f"3+3 decoder blocks, vocab {cfg['output_dim']}), beam {args.beam}, "
                                   f"ctc_weight {args.ctc_weight}, reverse_weight {args.reverse_weight}",
                       "chunks_per_launch": per_launch, "parallelism": f"chunk-shard x{world}",
                       "tokens_per_step": int(ntok)},
            "roofline": roof,
            "stage_ms_per_step": {k: round(v["ms"], 3) for k, v in stages.items()} if stages else None,
        }
        if world == 1 and args.cpu_baseline_chunks > 0:
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
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in the C stdout buffer: get it out first,
        print(json.dumps(out), flush=True)  # so that the JSON line is the LAST line of stdout


if __name__ == "__main__":
    main()
