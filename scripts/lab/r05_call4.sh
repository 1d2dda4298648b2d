#!/bin/bash
# Round 5, GPU call 4: conv_block.hip second form (weights in registers, swapped operand roles, swizzled chunks).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call4; mkdir -p $O
echo "== conv_block + streamed tests"
timeout 400 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block or streamed_convolutions or implicit_gemm_convolutions or projection_shortcut" 2>&1 | tail -25
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_stem\": [0-9.]*" | tr "\n" " "; echo; }
for rep in 1 2; do
  for blk in 0 1; do
    echo -n "diar RVD_CONV_BLOCK=$blk: "
    RVB_LAB=1 RVD_CONV_BLOCK=$blk timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_block$blk.json | pickd
  done
done
