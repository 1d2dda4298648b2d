#!/bin/bash
# Round 4, GPU call 9: how many workgroups the multi-workgroup linkage loop should use -- alone, and underneath the ASR encoder
# of the joint pipeline (where its persistent workgroups compete with the GEMMs for CUs and for the XCD's L2).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call9; mkdir -p $O
RVD_LINKAGE_MB=1 RVD_LINKAGE_G=8 timeout 200 python -m pytest tests/test_diar_gpu.py -q -m gpu -x -k "sixteen" 2>&1 | grep -v "^shader\|^linkage n=" | tail -2
for g in 4 8 16; do
  echo "RVD_LINKAGE_G=$g"; RVD_LINKAGE_MB=1 RVD_LINKAGE_G=$g timeout 120 python scripts/linkage_bench.py 2>&1 | tail -2
  RVD_LINKAGE_MB=1 RVD_LINKAGE_G=$g timeout 200 python scripts/linkage_bench.py 27000 2>&1 | tail -1
done
pickj() { grep -o "\"ms_per_step\": [0-9.]*\|\"last_step_s\": {[^}]*}" | tr "\n" " "; echo; }
for cfg in "0 16" "1 16" "1 8" "1 4" "0 16" "1 16"; do
  set -- $cfg
  echo -n "joint 1 h, RVD_LINKAGE_MB=$1 G=$2: "
  RVD_LINKAGE_MB=$1 RVD_LINKAGE_G=$2 timeout 300 python bench_joint.py --hours 1 --steps 2 --warmup 1 2>/dev/null | pickj
done
for g in 8 16; do
  echo -n "diar 1 h, G=$g: "
  RVD_LINKAGE_G=$g timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
done
