#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s5
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 900 python -m pytest tests/test_fp8_gpu.py -q > $O/t_fp8.log 2>&1
tail -n 40 $O/t_fp8.log | cut -c1-300
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x -k "attention or golden or gemm2" > $O/t_attn.log 2>&1
tail -n 5 $O/t_attn.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
timeout 300 python bench.py $Q > $O/bench.log 2>&1; tail -n 1 $O/bench.log | cut -c1-1800
timeout 300 python bench.py $Q --dtype fp8 > $O/bench_fp8.log 2>&1; tail -n 1 $O/bench_fp8.log | cut -c1-2200
