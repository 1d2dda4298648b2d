#!/bin/bash
# round 6, call 17: the segmentation of the windows behind the first trunk pass on its own stream, underneath that pass (RVD_SEG_OVERLAP=1, default) against all windows first (0): segmentation tests + stage times -- the code of this experiment is not in the tree (profiles/r06_call17_*)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call17; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_diar_gpu.py tests/test_diar_pipeline_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
for X in 1 0 1 0 1 0; do
  RVD_SEG_OVERLAP=$X timeout 300 python bench_diar.py --steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | grep '^{' | tail -1 > $O/diar_$X.json
  python - <<PY
import json
d=json.load(open("$O/diar_$X.json"))
s=d["stage_ms_per_step"]
print("SEG_OVERLAP=$X ms/step", d["ms_per_step"], "pool_norm", s["pool_norm"], "sincnet_conv", s["sincnet_conv"], "sinc_conv", s["sinc_conv"], "host", d["host_s_last_step"])
PY
done 2>&1 | tee $O/ab.txt
