"""BASELINE config 4 shape: full diarization pipeline on synthetic audio.
python scripts/diar_pipeline_bench.py [hours] [dtype]"""
import sys, time, tempfile, os, json
import numpy as np
sys.path.insert(0, ".")
from reverb_amd import synth_diar as SD, diarization as D

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tmp = tempfile.mkdtemp()
pipe = D.Pipeline.from_pretrained(SD.write_pipeline_dir(os.path.join(tmp, "pipe")), dtype=dtype).to("cuda")
base = SD.synth_conversation(120.0)
pcm = np.tile(base, int(hours * 30 + 0.999))[: int(hours * 3600 * 16000)]
pcm = (pcm.astype(np.int32) + np.random.default_rng(7).integers(-3, 4, size=pcm.size)).clip(-32768, 32767).astype(np.int16)
wave = {"waveform": pcm, "sample_rate": 16000, "uri": "bench"}
for rep in range(2):
    pipe.engine.set_profiling(rep == 1); pipe.engine.reset_timings()
    t0 = time.time(); ann = pipe(wave); dt = time.time() - t0
    print(f"rep {rep}: {hours} h in {dt:.3f} s  RTFx {hours*3600/dt:.0f}  turns {len(ann)} speakers {len(ann.labels())}")
    print("   ", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipe.timings.items()}))
for k in ("sinc_conv", "pool_norm", "sincnet_conv", "lstm_inproj", "lstm_recurrence", "linear", "classifier", "d2h", "emb_fbank", "emb_cmn", "emb_stem", "emb_conv_32", "emb_conv_64", "emb_conv_128", "emb_conv_256", "emb_conv_s2_64", "emb_conv_s2_128", "emb_conv_s2_256", "emb_conv_sc", "emb_pool", "emb_linear"):
    ms, fl, n = pipe.engine.timing(k)
    print(f"  {k:16s} {ms:9.2f} ms  {n:5d} launches  {fl/ms/1e9 if ms else 0:8.1f} TFLOP/s")
