#!/bin/bash
# Round 4, GPU call 16: the full-tile epilogues with the store hazard fixed (s_nop 1) -- correctness (kernel tests twice, engine /
# fp8 / long-form parity, diarization) and the A/B: RVB_GEMM2_FLAGS 1024 = round-3 epilogue, 0 = default (residual tiles fast for
# K <= 2048 only), 2048 = residual tiles fast for every K.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call16; mkdir -p $O
echo "== unit tests"
for r in 1 2; do timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu 2>&1 | grep -a -E "passed|failed" | tail -1; done
timeout 400 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "streamed or implicit_gemm or fused_residual or embedding" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -4
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py tests/test_longform_gpu.py -q -m gpu 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -6
echo "== gemm_bench: default (0) vs round-3 epilogue (1024) vs residual tiles fast for every K (2048)"
timeout 200 python scripts/gemm_bench.py 0,-2 1024,-2 2048,-2 2>&1 | tee $O/gemm_bench_switches.txt
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"gemm_fp8\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 1024 0 2048 1024 0 2048; do
  echo -n "bf16 RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_bf16_f$f.json | pick
done
for f in 1024 0 2048; do
  echo -n "fp8  RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py --dtype fp8 $B 2>/dev/null | tee $O/bench_fp8_f$f.json | pick
done
D="--steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*" | tr "\n" " "; echo; }
for st in 0 1 0 1; do
  echo -n "diar RVD_CONV_STREAM=$st: "
  RVD_CONV_STREAM=$st timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_stream$st.json | pickd
done
