#!/bin/bash
# Round 4, GPU call 11: (a) the full-tile GEMM epilogues whose stores the waitcnt pass does not see (RVB_GEMM2_FLAGS bit 10 = off,
# i.e. the round-3 epilogue), bf16 and fp8; (b) the implicit-GEMM convolution epilogue with the residual ring + hidden stores;
# (c) conv_stream.hip for the 32-/64-channel stages (RVD_CONV_STREAM = 0 off | n = time axis of a row split over n workgroups);
# (d) windows per ResNet pass (RVD_EMB_BATCH).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call11; mkdir -p $O
echo "== unit tests"
timeout 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -4
timeout 400 python -m pytest tests/test_diar_gpu.py -q -m gpu -x -k "streamed or implicit_gemm or fused_residual or embedding" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -6
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_fp8_gpu.py -q -m gpu -x -k "bf16_engine_within or f32_engine_matches or fp8" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -4
echo "== gemm_bench: fast epilogue (flags 0) vs round-3 epilogue (flags 1024)"
timeout 200 python scripts/gemm_bench.py 0,-2 1024,-2 2>&1 | tee $O/gemm_bench_switches.txt
echo "== gemm_timeline (default flags)"
timeout 120 python scripts/gemm_timeline.py 2>&1 | tee $O/gemm_timeline.txt | grep -E "^==|epilogue|main loop|gap"
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"gemm_fp8\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 1024 0 1024 0; do
  echo -n "bf16 RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_bf16_f$f.json | pick
done
for f in 1024 0; do
  echo -n "fp8  RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py --dtype fp8 $B 2>/dev/null | tee $O/bench_fp8_f$f.json | pick
done
D="--steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_32\": [0-9.]*\|\"emb_conv_64\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*\|\"emb_conv_s2_[0-9]*\": [0-9.]*" | tr "\n" " "; echo; }
for st in 0 2 1 3 0 2; do
  echo -n "diar RVD_CONV_STREAM=$st: "
  RVD_CONV_STREAM=$st timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_stream$st.json | pickd
done
for nb in 96 48 384; do
  echo -n "diar RVD_EMB_BATCH=$nb RVD_CONV_STREAM=2: "
  RVD_EMB_BATCH=$nb RVD_CONV_STREAM=2 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_batch$nb.json | pickd
done
