#!/bin/bash
# Round 4, GPU call 10: fp8 saturation counters (test + no slowdown of the fp8 step), the CPU shard tests on a box WITH a GPU.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call10; mkdir -p $O
timeout 400 python -m pytest tests/test_fp8_gpu.py -q -m gpu -x -k "saturation or shared or policy_is or layernorm_to_fp8 or gemm" 2>&1 | tail -4
timeout 400 python -m pytest tests/test_shard_gloo.py tests/test_bench_contract.py -q -x 2>&1 | tail -2
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie --dtype fp8"
for i in 1 2; do timeout 150 python bench.py $B 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"gemm_fp8\": [0-9.]*\|\"rownorm\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; done
