#!/bin/bash
# round 6, call 18: LSTM recurrence with the input projection's columns in the kernel's read order (one 16-byte load per window and step
# instead of eight 2-byte ones): segmentation tests + stage times (compare lstm_recurrence with profiles/r06_call15_*: 6.25-6.29 ms)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call18; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_diar_gpu.py tests/test_diar_pipeline_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; grep -a "passed\|failed" $O/pytest.log | tail -2
for X in 1 2; do
  timeout 300 python bench_diar.py --steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | grep '^{' | tail -1 > $O/diar_$X.json
  python - <<PY
import json
d=json.load(open("$O/diar_$X.json"))
s=d["stage_ms_per_step"]
print("run $X ms/step", d["ms_per_step"], "lstm_recurrence", s["lstm_recurrence"], "lstm_inproj", s["lstm_inproj"], "pool_norm", s["pool_norm"], "sincnet_conv", s["sincnet_conv"], "segmentation host s", d["host_s_last_step"]["segmentation"])
PY
done 2>&1 | tee $O/ab.txt
