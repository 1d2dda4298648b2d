"""joint_decoding host threads A/B on one box: python scripts/joint_threads_ab.py   (light case: default length bonus; heavy: 8)"""
import os, subprocess, sys
code = r'''
import sys, time
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth
from reverb_amd.engine import Engine
hours, chunk = 1.0, 2051
cfg, sd = synth.calibrated_state_dict("r640", 0)
n_samples = int(hours * 3600 * 16000)
n_chunks = -(-(1 + (n_samples - 400) // 160) // chunk)
eng = Engine(cfg, sd, dtype="bf16", device=0, max_chunks=n_chunks, chunk_frames=chunk)
eng.upload_pcm(synth.synth_audio(hours * 3600, seed=1234))
nf = eng.fbank()
lens = np.full(n_chunks, chunk, np.int32); lens[-1] = nf - (n_chunks - 1) * chunk
for bonus in (0.5, 3.0, 8.0):
    best = 1e9
    for rep in range(3):
        eng.encode(None, lens, 4, first_chunk=0, T0=chunk, topk=6)
        t0 = time.time()
        res = eng.joint_decode(0.3, bonus)
        best = min(best, time.time() - t0)
    print("  bonus %.1f: %8.1f ms  tokens %d  rows %d" % (bonus, best * 1e3, sum(len(r.tokens) for r in res), eng.joint_stats()[0]), flush=True)
'''
for thr in ("1", "4", "16"):
    print("RVB_SEARCH_THREADS=" + thr, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RVB_SEARCH_THREADS=thr))
