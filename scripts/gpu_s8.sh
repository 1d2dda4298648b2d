#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s8
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
for sl in 32 48 64 88 112; do RVB_SLICE0=$sl timeout 300 python bench.py $Q > $O/bench_sl$sl.log 2>&1; tail -n 1 $O/bench_sl$sl.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slice0 $sl', d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step'])"; done
