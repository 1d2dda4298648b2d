#!/bin/bash
# round 6, call 21: pointwise_conv1 + GLU in the GEMM's epilogue (ACT_GLU; RVB_GLU_FUSE=1) against GEMM + gate-while-staging (0)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call21; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_longform_gpu.py -q -m gpu -x -k "glu or gemm or attention_with_keys or r640_chunk" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
for X in 1 0 1 0; do
  RVB_LAB=1 RVB_GLU_FUSE=$X RVB_BENCH_LONG=$O/long_$X.json timeout 300 python bench.py --steps 10 --warmup 3 $N > $O/bench_$X.log 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/long_$X.json"))
s=d["stage_ms_per_step"]
print("GLU_FUSE=$X ms/step", d["ms_per_step"], "glu_dwconv", s["glu_dwconv"], "gemm", s["gemm"], "rownorm", s["rownorm"], "frac", d["roofline"]["frac"], "tokens", d["config"]["tokens_per_step"])
PY
done 2>&1 | tee $O/ab.txt
