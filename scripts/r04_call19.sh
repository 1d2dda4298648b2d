#!/bin/bash
# Round 4, GPU call 19: the persistent loop without cross-tile prefetch (RVB_GEMM2_FLAGS bit 12)
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call19; mkdir -p $O
for r in 1 2; do timeout 200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "persistent or tuning_switches or ring" 2>&1 | grep -a -E "passed|failed" | tail -1; done
echo "== gemm_bench 0 vs 4096 vs 16"
timeout 200 python scripts/gemm_bench.py 0,-2 4096,-2 16,-2 2>&1 | tee $O/gemm_bench_switches.txt | tail -13
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 0 4096 0 4096; do
  echo -n "bf16 RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_bf16_f$f.json | pick
done
