#!/bin/bash
# round 6, call 3: the folded rel-pos attention with K' = k + p written by the qkv GEMM (default) against the two-product form
# (RVB_ATTN_PREFOLD=0, lab library): stage table + headline step of both; the parity tests of the r640 model; then the GPU suite
# and the driver's bench line.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call3; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for F in 1 0 1 0; do
  RVB_LAB=1 RVB_ATTN_PREFOLD=$F RVB_BENCH_LONG=$O/long_prefold$F.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_prefold$F.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_prefold$F.json"))
s=d["stage_ms_per_step"]
print("prefold=$F ms/step", d["ms_per_step"], "attention", s["attention"], "gemm", s["gemm"], "rownorm", s["rownorm"], "frac", d["roofline"]["frac"])
PY
done 2>&1 | tee $O/prefold_ab.txt
timeout 900 python -m pytest tests/test_longform_gpu.py tests/test_engine_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/pytest_r640.log 2>&1; tail -n 4 $O/pytest_r640.log
timeout 1500 python -m pytest tests -q -m gpu -x -rs > $O/pytest_gpu.log 2>&1; tail -n 5 $O/pytest_gpu.log
cp gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r640.log 2> $O/bench_r640.err; tail -c 1900 $O/bench_r640.log
cp gpurun_out/bench_long.json $O/ 2>/dev/null
ls $O
