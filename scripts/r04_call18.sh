#!/bin/bash
# Round 4, GPU call 18: start stagger of the fp32 + residual GEMMs (RVB_GEMM2_STAGGER: 0 off, 1 = half a tile by estimate, n = n us)
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call18; mkdir -p $O
RVB_GEMM2_STAGGER=1 timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "tuning_switches or ring or persistent" 2>&1 | grep -a -E "passed|failed" | tail -1
for st in 0 1; do echo "== gemm_bench RVB_GEMM2_STAGGER=$st"; RVB_GEMM2_STAGGER=$st timeout 200 python scripts/gemm_bench.py 0,-2 2>&1 | grep -E "ffn2|out/pw2|sum"; done
echo "== timeline with stagger"; RVB_GEMM2_STAGGER=1 timeout 120 python scripts/gemm_timeline.py 2>&1 | grep -A7 "^== out/pw2\|^== ffn2" | grep -E "^==|epilogue|main loop|launch"
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for st in 0 1 10 30 0 1; do
  echo -n "bf16 RVB_GEMM2_STAGGER=$st: "
  RVB_GEMM2_STAGGER=$st timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_st$st.json | pick
done
