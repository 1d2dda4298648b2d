#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s9
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_diar_pipeline_gpu.py -q -x -k "bench_workload or pipeline or joint or gemm" > $O/t.log 2>&1; tail -n 5 $O/t.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null; cat $O/parity_metrics.jsonl | cut -c1-400
timeout 600 python bench_joint.py --steps 3 --warmup 1 > $O/bench_joint.log 2>&1; tail -n 1 $O/bench_joint.log | cut -c1-1500
timeout 600 python bench_joint.py --steps 3 --warmup 1 --dtype bf16 > $O/bench_joint_bf16.log 2>&1; tail -n 1 $O/bench_joint_bf16.log | cut -c1-400
