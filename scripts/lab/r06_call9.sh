#!/bin/bash
# round 6, call 9: the folded attention with the head's per-key constants in LDS once per workgroup (RVB_ATTN_PREFOLD=1) against the two-product form (0)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call9; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_longform_gpu.py tests/test_streaming_gpu.py tests/test_causal_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for X in 1 0 1 0; do
  RVB_LAB=1 RVB_ATTN_PREFOLD=$X RVB_BENCH_LONG=$O/long_$X.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_$X.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_$X.json"))
s=d["stage_ms_per_step"]
print("PREFOLD=$X ms/step", d["ms_per_step"], "attention", s["attention"], "gemm", s["gemm"], "frac", d["roofline"]["frac"], "tokens", d["config"]["tokens_per_step"])
PY
done 2>&1 | tee $O/ab.txt
ls $O
