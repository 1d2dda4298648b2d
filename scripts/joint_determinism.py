import sys, os, time
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth
from reverb_amd.engine import Engine
hours = 0.25
chunk = 2051
cfg, sd = synth.calibrated_state_dict("r640", 0)
n_samples = int(hours * 3600 * 16000)
n_chunks = -(-(1 + (n_samples - 400) // 160) // chunk)
eng = Engine(cfg, sd, dtype="bf16", device=0, max_chunks=n_chunks, chunk_frames=chunk)
eng.upload_pcm(synth.synth_audio(hours * 3600, seed=1234))
nf = eng.fbank()
lens = np.full(n_chunks, chunk, np.int32); lens[-1] = nf - (n_chunks - 1) * chunk
outs = []
for rep in range(3):
    eng.encode(None, lens, 4, first_chunk=0, T0=chunk, topk=6)
    t0 = time.time()
    res = eng.joint_decode(0.3, 8.0)
    outs.append([tuple(r.tokens) for r in res])
    print("rep", rep, "tokens", sum(len(r.tokens) for r in res), "ms", 1e3 * (time.time() - t0), flush=True)
print("identical:", outs[0] == outs[1] == outs[2])
