#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 60 scripts/micro/tr_probe > $O/tr_probe.txt 2>&1
head -n 70 $O/tr_probe.txt | tail -n 66 | awk 'NR%4==1' | head -20
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 1500 python -m pytest tests -q -x -m gpu -k "kernels or engine or longform or edge" > $O/t.log 2>&1
tail -n 12 $O/t.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
timeout 300 python bench.py $Q > $O/bench.log 2>&1; tail -n 1 $O/bench.log | cut -c1-1800
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0 --no-profile > $O/prof_stdout.log 2>&1
cd $R
find $O/prof -name "*kernel_trace.csv" -delete
F=$(find $O/prof -name "*kernel_stats.csv" | head -n 1)
[ -n "$F" ] && cut -c1-200 $F | head -30
