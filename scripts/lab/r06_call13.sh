#!/bin/bash
# round 6, call 13 (3 without the index list = no scratch; 4 = 16 waves, 512 queries per workgroup): the prefolded attention with two 16-query fragments per wave (RVB_ATTN_MF = 1: 4 waves / 128 queries per
# workgroup, 2: 8 waves / 256 queries, 3: as 2 in 128 VGPRs) against one fragment per wave (0, the committed kernel)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call13; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
for X in 3 4; do
  RVB_LAB=1 RVB_ATTN_MF=$X timeout 600 python -m pytest tests/test_longform_gpu.py tests/test_engine_gpu.py -q -m gpu -x > $O/pytest_$X.log 2>&1; echo "MF=$X: $(tail -n 1 $O/pytest_$X.log)" | tee -a $O/ab.txt
done
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for X in 0 3 4 0 3 4; do
  RVB_LAB=1 RVB_ATTN_MF=$X RVB_BENCH_LONG=$O/long_$X.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_$X.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_$X.json"))
s=d["stage_ms_per_step"]
print("ATTN_MF=$X ms/step", d["ms_per_step"], "attention", s["attention"], "gemm", s["gemm"], "frac", d["roofline"]["frac"], "tokens", d["config"]["tokens_per_step"])
PY
done 2>&1 | tee -a $O/ab.txt
