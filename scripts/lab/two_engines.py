#!/usr/bin/env python3
"""Lab: the bench hour as TWO half-hour engines on two HIP streams, driven by two host threads, against one engine with the
whole hour (the committed bench step).  Question: do the tile-count tails of the N = 1024 GEMMs (1 408 tiles on 256 CUs = 5.5
rounds) and the lock-step epilogue bursts fill up when a second stream has work for the idle CUs?
Usage: python scripts/lab/two_engines.py [--steps 10] [--warmup 3] [--lanes 2]"""
import argparse, json, threading, time
import torch
from reverb_amd import synth
from reverb_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--model", default="r640")
args = ap.parse_args()
chunk, beam, ctcw = 2051, 10, 0.1
cfg, sd = synth.calibrated_state_dict(args.model, 0)
modes = ["attention_rescoring"]


def make(seconds, seed):
    n_samples = int(round(seconds * 16000))
    n_frames = 1 + (n_samples - 400) // 160
    n_chunks = -(-n_frames // chunk)
    e = Engine(cfg, sd, dtype="bf16", device=0, max_chunks=n_chunks, chunk_frames=chunk)
    pcm = e.pinned_pcm(n_samples)
    pcm[:] = synth.synth_audio(seconds, seed=seed)
    e.upload_pcm(pcm)
    return e, n_chunks


def one(e):
    nf = e.fbank()
    h = e.decode_resident(nf, modes, chunk, beam, ctcw, 0.0)["attention_rescoring"]
    return sum(len(x.tokens) for x in h)


def run(engines, steps):
    tok = [0] * len(engines)
    def work(i):
        for _ in range(steps):
            tok[i] = one(engines[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(engines))]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, sum(tok)


out = {}
whole, nc = make(3600.0, 1234)
run([whole], args.warmup)
out["one_engine_ms"] = [round(run([whole], args.steps)[0], 2) for _ in range(2)]
out["one_engine_chunks"] = nc
del whole
torch.cuda.empty_cache()
L = args.lanes
parts = [make(3600.0 / L, 1234 + i) for i in range(L)]
engs = [p[0] for p in parts]
run(engs, args.warmup)
out["lanes"] = L
out["lanes_chunks"] = [p[1] for p in parts]
out["lanes_ms"] = [round(run(engs, args.steps)[0], 2) for _ in range(2)]
out["one_lane_alone_ms"] = round(run(engs[:1], args.steps)[0], 2)
print(json.dumps(out))
