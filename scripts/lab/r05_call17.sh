#!/bin/bash
# Round 5, GPU call 17: conv_s2.hip (stride-2 convolution + projection shortcut of the block that opens the 64-channel stage in one
# pass) against the two launches of the direct kernel (RVD_CONV_S2SC=0).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call17; mkdir -p $O
timeout 900 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "stride2 or fused_basic_block or row64 or implicit_gemm or shortcut or streamed or embedding" 2>&1 | tail -8
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_s2_64\": [0-9.]*\|\"emb_conv_sc\": [0-9.]*\|\"emb_conv_64\": [0-9.]*" | tr "\n" " "; echo; }
run() { echo -n "diar $1: "; env RVB_LAB=1 $1 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_$2.json | pickd; }
for rep in 1 2; do
  run "RVD_X=0" s2sc
  run "RVD_CONV_S2SC=0" two_launches
done
