#!/bin/bash
# Round 5, GPU call 15: column-strip forms of conv_block / conv_row64 (a pixel fragment read once per input row and used for three
# output rows) and the 512-pixel implicit-GEMM tile of the 128-channel stage, each against the form it replaces.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call15; mkdir -p $O
timeout 600 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block or row64 or implicit_gemm or shortcut" 2>&1 | tail -8
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_s2_128\": [0-9.]*" | tr "\n" " "; echo; }
run() { echo -n "diar $1: "; env RVB_LAB=1 $1 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_$2.json | pickd; }
for rep in 1 2; do
  run "RVD_X=0" default
  run "RVD_CONV_BLOCK=1" block_rows
  run "RVD_CONV_ROW64=2" row64_pairs
  run "RVD_IGEMM_BM=256" igemm_256
done
