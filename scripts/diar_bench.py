"""Stage timings of the diarization networks on synthetic audio: python scripts/diar_bench.py [hours] [dtype]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth_diar as SD
from reverb_amd.diar_engine import DiarEngine

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg = SD.make_diar_config()
eng = DiarEngine(cfg, SD.make_segmentation_sd(cfg, 0), dtype=dtype)
base = SD.synth_conversation(60.0)
pcm = np.tile(base, int(hours * 60 + 0.999))[: int(hours * 3600 * 16000)]
for rep in range(2):
    eng.set_profiling(rep == 1)
    eng.reset_timings()
    t0 = time.time()
    W = eng.upload(pcm)
    t1 = time.time()
    lp = eng.segment(batch=batch)
    t2 = time.time()
    print(f"rep {rep}: windows {W} upload {1e3*(t1-t0):.1f} ms segment {1e3*(t2-t1):.1f} ms  RTFx {hours*3600/(t2-t0):.0f}")
for k in ("h2d", "pcm_to_float", "sinc_conv", "window_stats", "pool_norm", "sincnet_conv", "lstm_inproj", "lstm_recurrence", "linear", "classifier", "d2h"):
    ms, fl, n = eng.timing(k)
    print(f"  {k:16s} {ms:9.2f} ms  {n:4d} launches  {fl/ms/1e9 if ms else 0:8.1f} TFLOP/s")
print("classes", np.bincount(lp.argmax(-1).ravel(), minlength=7))
