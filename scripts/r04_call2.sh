#!/bin/bash
# Round 4, GPU call 2: static GEMM tail + K serpentine (RVB_GEMM2_FLAGS bit 7), XCD-aware attention block order (RVB_ATTN_PLAIN=1 =
# the old order), double-buffered PCM upload (pcie_inclusive), stride-2 implicit-GEMM convolutions, and config 5's 3-hour shape.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call2; mkdir -p $O
echo "== unit tests"
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | grep -a -E "passed|failed|Error" | tail -3
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_diar_gpu.py -q -m gpu -x -k "double_buffered or implicit_gemm or f32_engine_matches or bf16_engine_within or embedding" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -5
echo "== gemm_bench: flags 0 vs 128 (K serpentine)"
timeout 200 python scripts/gemm_bench.py 0,-2 128,-2 2>&1 | tee $O/gemm_bench_switches.txt
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"rownorm\": [0-9.]*\|\"glu_dwconv\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for cfg in "0 1" "0 0" "128 0" "0 1" "128 0"; do
  set -- $cfg
  echo -n "RVB_GEMM2_FLAGS=$1 RVB_ATTN_PLAIN=$2: "
  RVB_GEMM2_FLAGS=$1 RVB_ATTN_PLAIN=$2 timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_f$1_p$2.json | pick
done
echo "== diarization: stride-2 convolutions direct (RVD_CONV_IGEMM=1) vs implicit GEMM (default)"
for ig in 1 2; do
  echo -n "RVD_CONV_IGEMM=$ig: "
  RVD_CONV_IGEMM=$ig timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar_ig$ig.json | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"emb_conv[a-z0-9_]*\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*\|\"clustering[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
done
echo "== config 5's audio length: bench_joint --hours 3 on one GPU"
timeout 400 python bench_joint.py --hours 3 --steps 1 --warmup 1 2>$O/joint3h.err | tee $O/joint_3h.json | cut -c1-1500
