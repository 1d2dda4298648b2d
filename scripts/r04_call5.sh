#!/bin/bash
# Round 4, GPU call 5: transposed-accumulator epilogue (RVB_GEMM2_FLAGS bit 9), linkage with eager validation of near-top stale
# rows, joint_decoding with pre-beams beyond 16 and a blank penalty.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call5; mkdir -p $O
echo "== tests"
RVB_GEMM2_FLAGS=512 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_joint_gpu.py tests/test_diar_gpu.py -q -m gpu -x -k "joint or linkage" 2>&1 | grep -v "^shader\|^linkage n=" | tail -5
echo "== gemm_bench: flags 0 vs 512 (transposed accumulators, direct stores)"
timeout 200 python scripts/gemm_bench.py 0,-2 512,-2 2>&1 | tee $O/gemm_bench_switches.txt
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 0 512 0 512; do
  echo -n "RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_f$f.json | pick
done
echo "== timeline of ffn1 / qkv with and without"
for f in 0 512; do echo "flags $f"; RVB_GEMM2_FLAGS=$f timeout 100 python scripts/gemm_timeline.py 2>&1 | grep -A7 "^== ffn1\|^== qkv\|^== plain" | grep "==\|prologue\|epilogue\|main loop\|gap" ; done
echo "== linkage bench"
RVD_LINKAGE_MB=1 timeout 120 python scripts/linkage_bench.py 2>&1 | tail -3
RVD_LINKAGE_MB=1 timeout 200 python scripts/linkage_bench.py 27000 2>&1 | tail -1
echo "== diarization, 1 h"
timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar.json | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
