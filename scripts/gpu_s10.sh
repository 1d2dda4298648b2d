#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s10
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_edge_cases_gpu.py -q -x > $O/t.log 2>&1; tail -n 4 $O/t.log
Q="--steps 2 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
RVB_FORCE_DIST=1 timeout 300 python bench.py $Q > $O/bench_dist.log 2>&1; tail -n 1 $O/bench_dist.log | cut -c1-900
RVB_FORCE_DIST=1 RVB_COMM=cabi timeout 300 python bench.py $Q > $O/bench_cabi.log 2>&1; tail -n 1 $O/bench_cabi.log | cut -c1-900
