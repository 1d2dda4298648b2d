#!/bin/bash
# Round 5, GPU call 14: conv_row64 narrow-tile / two-buffer lab variant (RVD_CONV_ROW64=2) against the default (1) and the direct kernel (0).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call14; mkdir -p $O
timeout 400 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "row64" 2>&1 | tail -8
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_64\": [0-9.]*" | tr "\n" " "; echo; }
for rep in 1 2; do
  for r in 0 1 2; do
    echo -n "diar RVD_CONV_ROW64=$r: "
    RVB_LAB=1 RVD_CONV_ROW64=$r timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_row64_$r.json | pickd
  done
done
