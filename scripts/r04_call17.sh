#!/bin/bash
# Round 4, GPU call 17: (a) the persistent form of the GEMM (flag bit 4) again, now that its first K step no longer waits for the
# previous tile's stores and the epilogue is the fast one; (b) windows per ResNet trunk pass.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call17; mkdir -p $O
for r in 1 2 3; do timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "persistent or tuning_switches or ring" 2>&1 | grep -a -E "passed|failed" | tail -1; done
echo "== gemm_bench 0 vs 16"
timeout 200 python scripts/gemm_bench.py 0,-2 16,-2 2>&1 | tee $O/gemm_bench_switches.txt | tail -13
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 0 16 0 16; do
  echo -n "bf16 RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_bf16_f$f.json | pick
done
D="--steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*" | tr "\n" " "; echo; }
for nb in 192 384 768 192 384; do
  echo -n "diar RVD_EMB_BATCH=$nb: "
  RVD_EMB_BATCH=$nb timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_batch$nb.json | pickd
done
