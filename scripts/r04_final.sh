#!/bin/bash
# Round 4, last GPU call: the GPU test suite, the default bench command and its rocprofv3 kernel stats at the final tree (start
# stagger of the residual GEMMs on; everything else as in the r04c run).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_r640.log 2> $O/bench_r640.err
tail -n 1 $O/bench_r640.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_asr -- python $R/bench.py --steps 2 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0 > $O/prof_asr_stdout.log 2>&1 < /dev/null
find $O -name "*kernel_trace.csv" -delete
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 1100 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 3 $O/pytest_gpu.log
cp $R/gpurun_out/parity_metrics.jsonl $O/parity_metrics.jsonl 2>/dev/null
