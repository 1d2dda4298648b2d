#!/bin/bash
# round 6, call 15: SincNet conv layers 2 and 3 on the thin-GEMM kernel conv1d5 (RVD_CONV1D5=1, default) against the generic GEMM (0): segmentation tests + stage times
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call15; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_diar_gpu.py tests/test_diar_pipeline_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for X in 1 0 1 0; do
  RVB_LAB=1 RVD_CONV1D5=$X timeout 300 python bench_diar.py --steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | grep '^{' | tail -1 > $O/diar_$X.json
  python - <<PY
import json
d=json.load(open("$O/diar_$X.json"))
s=d["stage_ms_per_step"]
print("CONV1D5=$X ms/step", d["ms_per_step"], "pool_norm", s["pool_norm"], "sincnet_conv", s["sincnet_conv"], "sinc_conv", s["sinc_conv"], "segmentation host s", d["host_s_last_step"]["segmentation"])
PY
done 2>&1 | tee $O/ab.txt
