"""Timing of the widened decoding modes on the bench workload: python scripts/modes_bench.py [hours] [model]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth
from reverb_amd.engine import Engine

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
model = sys.argv[2] if len(sys.argv) > 2 else "r640"
chunk = 2051
cfg, sd = synth.calibrated_state_dict(model, 0)
cfg["encoder_conf"]["use_dynamic_chunk"] = True
n_samples = int(hours * 3600 * 16000)
n_chunks = -(-(1 + (n_samples - 400) // 160) // chunk)
eng = Engine(cfg, sd, dtype="bf16", device=0, max_chunks=n_chunks, chunk_frames=chunk)
eng.upload_pcm(synth.synth_audio(hours * 3600, seed=1234))


def run(name, modes, cs=-1, left=-1):
    eng.apply_decoding_chunk(cs, left)
    for rep in range(2):
        eng.reset_timings(); eng.set_profiling(rep == 1)
        t0 = time.time()
        nf = eng.fbank()
        res = eng.decode_resident(nf, modes, chunk, 10, 0.1, 0.0)
        dt = time.time() - t0
    att = eng.timing("attention")
    ntok = sum(len(r.tokens) for r in res[modes[0]])
    print(f"{name:46s} {dt*1e3:9.1f} ms  RTFx {hours*3600/dt:8.0f}  tokens {ntok}  attention kernels {att['ms']:.1f} ms")


run("attention_rescoring (full context)", ["attention_rescoring"])
run("attention_rescoring, decoding_chunk 16 / left 4", ["attention_rescoring"], 16, 4)
run("attention_rescoring, decoding_chunk 16 / all left", ["attention_rescoring"], 16, -1)
run("attention (autoregressive beam 10)", ["attention"])
