#!/bin/bash
# Round 5, GPU call 18: conv_flat.hip (stride-1 3x3 convolutions of the 128- / 256-channel stages over the flat bordered index, the
# tile's pixels resident in LDS for all nine taps) against the implicit GEMM (RVD_CONV_FLAT=0).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call18; mkdir -p $O
timeout 600 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "flat_conv or implicit_gemm" 2>&1 | tail -12
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*" | tr "\n" " "; echo; }
run() { echo -n "diar $1: "; env RVB_LAB=1 $1 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_$2.json | pickd; }
for rep in 1 2; do
  run "RVD_X=0" flat
  run "RVD_CONV_FLAT=0" igemm
done
