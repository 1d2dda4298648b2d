#!/bin/bash
# GPU session 2 (round 2): full GPU suite after the trie rescoring, bench, gemm2 switches end to end
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 1500 python -m pytest tests -q -x -m gpu > $O/t_all.log 2>&1
tail -n 25 $O/t_all.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 3 --warmup 1 --no-diarization > $O/bench.log 2> $O/bench.err
tail -n 1 $O/bench.log | cut -c1-2500
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
RVB_GEMM2_GROUP_M=8 timeout 300 python bench.py $Q > $O/bench_g8.log 2>&1; tail -n 1 $O/bench_g8.log | cut -c1-330
RVB_GEMM2_FLAGS=2 timeout 300 python bench.py $Q > $O/bench_prio.log 2>&1; tail -n 1 $O/bench_prio.log | cut -c1-330
RVB_GEMM2_FLAGS=1 timeout 300 python bench.py $Q > $O/bench_mma32.log 2>&1; tail -n 1 $O/bench_mma32.log | cut -c1-330
