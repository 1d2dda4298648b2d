#!/bin/bash
# Round 5, GPU call 10: conv_row64.hip (64-channel stage) against the direct kernel; conv_block in its final (4-wave) form.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call10; mkdir -p $O
timeout 500 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block or row64 or streamed_convolutions or embedding" 2>&1 | tail -15
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*" | tr "\n" " "; echo; }
for rep in 1 2; do
  for r in 0 1; do
    echo -n "diar RVD_CONV_ROW64=$r: "
    RVB_LAB=1 RVD_CONV_ROW64=$r timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_row64_$r.json | pickd
  done
done
