#!/bin/bash
# Round 4, GPU call 8: the multi-workgroup linkage loop with the tagged exchange (publishing is the barrier).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call8; mkdir -p $O
timeout 300 python -m pytest tests/test_diar_gpu.py -q -m gpu -x -k "linkage" 2>&1 | grep -v "^shader\|^linkage n=" | tail -4
for mb in 1; do
  echo "RVD_LINKAGE_MB=$mb"; RVD_LINKAGE_MB=$mb timeout 120 python scripts/linkage_bench.py 2>&1 | tail -3
  RVD_LINKAGE_MB=$mb timeout 200 python scripts/linkage_bench.py 27000 2>&1 | tail -1
done
timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar.json | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
timeout 400 python bench_joint.py --hours 3 --steps 1 --warmup 1 2>/dev/null | tee $O/joint_3h.json | grep -o "\"ms_per_step\": [0-9.]*\|\"last_step_s\": {[^}]*}" | tr "\n" " "; echo
