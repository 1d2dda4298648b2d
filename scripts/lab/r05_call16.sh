#!/bin/bash
# Round 5, GPU call 16: band form of conv_block (a workgroup walks down the rows of a 60-frame band, patch and mid rows in rings:
# no row of either convolution computed twice) against the column strips on 4-row tiles (RVD_CONV_BLOCK=2) and the row form (1).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call16; mkdir -p $O
timeout 600 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block" 2>&1 | tail -8
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*" | tr "\n" " "; echo; }
run() { echo -n "diar $1: "; env RVB_LAB=1 $1 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_$2.json | pickd; }
for rep in 1 2; do
  run "RVD_X=0" band
  run "RVD_CONV_BLOCK=2" strips
  run "RVD_CONV_BLOCK=1" rows
done
