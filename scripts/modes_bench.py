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


def run_joint(beam, ctc_weight, bonus):
    for rep in range(2):
        t0 = time.time()
        nf = eng.fbank()
        lens = np.full(n_chunks, chunk, np.int32); lens[-1] = nf - (n_chunks - 1) * chunk
        eng.encode(None, lens, beam, first_chunk=0, T0=chunk, topk=int(1.5 * beam))
        t1 = time.time()
        res = eng.joint_decode(ctc_weight, bonus)
        dt = time.time() - t0
    rows, steps = eng.joint_stats()
    ntok = sum(len(r.tokens) for r in res)
    print(f"joint_decoding beam {beam} ctc {ctc_weight} bonus {bonus:4.1f}            {dt*1e3:9.1f} ms  RTFx {hours*3600/dt:8.0f}  tokens {ntok}  "
          f"search {1e3*(time.time()-t1):.1f} ms, {rows} decoder rows in {steps} batched steps")


eng.apply_decoding_chunk(-1, -1)
run_joint(4, 0.3, 0.5)
run_joint(4, 0.3, 8.0)
run_joint(10, 0.5, 8.0)
