#!/bin/bash
# Round 5, GPU call 9: conv_block four-wave / two-workgroups-per-CU lab variant (RVD_CONV_BLOCK=2) against the default (1) and two launches (0).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call9; mkdir -p $O
timeout 400 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block" 2>&1 | tail -15
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*" | tr "\n" " "; echo; }
for rep in 1 2; do
  for blk in 0 1 2; do
    echo -n "diar RVD_CONV_BLOCK=$blk: "
    RVB_LAB=1 RVD_CONV_BLOCK=$blk timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_block$blk.json | pickd
  done
done
