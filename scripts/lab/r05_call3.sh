#!/bin/bash
# Round 5, GPU call 3: the fused 32-channel BasicBlock (conv_block.hip) -- tests first, then the whole GPU suite, then A/B on the hour.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call3; mkdir -p $O
rm -f gpurun_out/parity_metrics.jsonl
echo "== conv_block tests"
timeout 300 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "fused_basic_block" 2>&1 | tail -25
echo "== pytest -m gpu (whole suite)"
timeout 1200 python -m pytest tests/ -q -m gpu -rs 2>&1 | tee $O/pytest_gpu.txt | tail -30
cp gpurun_out/parity_metrics.jsonl $O/parity_metrics.jsonl 2>/dev/null
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_stem\": [0-9.]*" | tr "\n" " "; echo; }
for rep in 1 2; do
  for blk in 0 1; do
    echo -n "diar RVD_CONV_BLOCK=$blk: "
    RVB_LAB=1 RVD_CONV_BLOCK=$blk timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_block$blk.json | pickd
  done
done
echo -n "diar product library (default): "
timeout 200 python bench_diar.py --steps 3 --warmup 1 --cpu-baseline-windows 0 2>/dev/null | tee $O/diar_default.json | pickd
