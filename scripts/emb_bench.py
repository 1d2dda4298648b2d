"""Timing of the embedding network: python scripts/emb_bench.py [windows] [dtype]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth_diar as SD
from reverb_amd.diar_engine import DiarEngine

W = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = SD.make_diar_config()
eng = DiarEngine(cfg, SD.make_segmentation_sd(cfg, 0), SD.make_embedding_sd(cfg, 0), dtype=dtype)
pcm = np.tile(SD.synth_conversation(60.0), (W + 9) // 60 + 2)[: (W + 9) * 16000]
nw = eng.upload(pcm)
wins = np.repeat(np.arange(W, dtype=np.int64), 3)
rng = np.random.default_rng(0)
masks = (rng.random((3 * W, 589)) < 0.5).astype(np.float32)
for rep in range(2):
    eng.set_profiling(rep == 1)
    eng.reset_timings()
    t0 = time.time()
    emb = eng.embed(wins, masks)
    dt = time.time() - t0
    print(f"rep {rep}: {W} windows, {3*W} embeddings in {1e3*dt:.1f} ms = {1e3*dt/W:.3f} ms/window -> 1 h (3591 windows): {dt/W*3591:.2f} s")
for k in ("emb_cmn", "emb_stem", "emb_conv_32", "emb_conv_64", "emb_conv_128", "emb_conv_256", "emb_conv_s2_64", "emb_conv_s2_128", "emb_conv_s2_256", "emb_conv_sc", "emb_pool", "emb_linear"):
    ms, fl, n = eng.timing(k)
    print(f"  {k:12s} {ms:9.2f} ms  {n:5d} launches  {fl/ms/1e9 if ms else 0:8.1f} TFLOP/s")
print("emb mean abs", float(np.abs(emb).mean()))
