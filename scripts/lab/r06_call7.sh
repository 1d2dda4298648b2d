#!/bin/bash
# round 6, call 7: the two-phase K step (gemm2p_kernel PH2, lab flag bit 14 = 16384): correctness on every epilogue path, kernel
# benchmark against the four-phase loop, and the headline step with it.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call7; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm" > $O/pytest_gemm.log 2>&1; tail -n 5 $O/pytest_gemm.log
RVB_LAB=1 timeout 600 python scripts/gemm_bench.py 0,-2 16384,-2 0,-2 16384,-2 > $O/gemm_bench_ph2.txt 2>&1; tail -n 14 $O/gemm_bench_ph2.txt
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for X in 0 16384 0 16384; do
  RVB_LAB=1 RVB_GEMM2_FLAGS=$X RVB_BENCH_LONG=$O/long_$X.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_$X.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_$X.json"))
s=d["stage_ms_per_step"]
print("FLAGS=$X ms/step", d["ms_per_step"], "gemm", s["gemm"], "attention", s["attention"], "frac", d["roofline"]["frac"], "tokens", d["config"]["tokens_per_step"])
PY
done 2>&1 | tee $O/ab.txt
ls $O
