#!/bin/bash
# round 6, call 4: (a) the persistent forms of the phase loop re-measured at this tree ("re-measure it on the new
# loop"): kernel benchmark + the headline step with RVB_GEMM2_FLAGS = 16 (cross-tile prefetch) and 4096 (persistent loop only);
# (b) three attention workgroups per CU (RVB_ATTN_OCC=3); (c) the MP3 tests incl. transcribe("x.mp3") on the GPU.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call4; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
RVB_LAB=1 timeout 600 python scripts/gemm_bench.py 0,-2 16,-2 4096,-2 0,-2 16,-2 4096,-2 > $O/gemm_bench_persistent.txt 2>&1; tail -n 14 $O/gemm_bench_persistent.txt
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for V in "FLAGS=0" "FLAGS=16" "FLAGS=4096" "OCC=3" "FLAGS=0" "OCC=3"; do
  K=${V%%=*}; X=${V##*=}
  if [ $K = FLAGS ]; then export RVB_GEMM2_FLAGS=$X; unset RVB_ATTN_OCC; else export RVB_ATTN_OCC=$X; unset RVB_GEMM2_FLAGS; fi
  RVB_LAB=1 RVB_BENCH_LONG=$O/long_$K$X.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_$K$X.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_$K$X.json"))
s=d["stage_ms_per_step"]
print("$V ms/step", d["ms_per_step"], "attention", s["attention"], "gemm", s["gemm"], "frac", d["roofline"]["frac"], "tokens", d["config"]["tokens_per_step"])
PY
done 2>&1 | tee $O/ab.txt
unset RVB_GEMM2_FLAGS RVB_ATTN_OCC
timeout 900 python -m pytest tests/test_mp3.py tests/test_audio_decode.py -q -rs > $O/pytest_mp3.log 2>&1; tail -n 4 $O/pytest_mp3.log
ls $O
