"""Where the replicated host part of a 3 h diarization goes on the GPU box's CPU: cProfile of SpeakerDiarization.finish()."""
import cProfile, pstats, sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from reverb_amd import diarization as D, synth_diar as SD
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
n = int(hours * 3600 * 16000)
dcfg = SD.make_diar_config()
pipe = D.SpeakerDiarization(dcfg, SD.make_segmentation_sd(dcfg, 0), SD.make_embedding_sd(dcfg, 0), None, dtype="bf16").to(0)
base = SD.synth_conversation(120.0)
pcm = np.tile(base, n // len(base) + 1)[:n]
pcm = (pcm.astype(np.int32) + np.random.default_rng(7).integers(-3, 4, size=n)).clip(-32768, 32767).astype(np.int16)
classes, emb = pipe.networks(pcm)
for i in range(3):
    t0 = time.perf_counter(); ann = pipe.finish(classes.copy(), emb, "x"); t1 = time.perf_counter()
    print("finish", round(t1 - t0, 4), {k: round(v, 4) for k, v in pipe.timings.items() if k in ("clustering", "reconstruction")}, len(ann))
c2 = classes.copy()
cProfile.run("pipe.finish(c2, emb, 'x')", "/tmp/prof_finish")
pstats.Stats("/tmp/prof_finish").sort_stats("tottime").print_stats(18)
