import sys, time, numpy as np
from reverb_amd import synth_diar as SD, diarization as D
from reverb_amd.diar_engine import DiarEngine
cfg = SD.make_diar_config()
eng = DiarEngine(cfg, SD.make_segmentation_sd(cfg, 0), dtype="bf16")
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9200
for name, K, noise in ((("4 clusters", 4, 0.3),) if N > 12000 else (("2 tight clusters", 2, 0.05), ("4 clusters", 4, 0.3), ("uniform", 1, 1.0))):
    n, d = N, 256
    c = rng.standard_normal((K, d))
    X = c[rng.integers(K, size=n)] + noise * rng.standard_normal((n, d))
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    eng.reset_timings(); eng.set_profiling(True)
    t = time.time(); Z = eng.centroid_linkage(X); dt = time.time() - t
    ms, retries, _ = eng.timing("linkage")
    print(f"{name:18s} n={n} wall {dt*1e3:.1f} ms  gpu {ms:.1f} ms  retries {retries:.0f} ({retries/(n-1):.2f}/merge)")
