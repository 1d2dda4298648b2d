#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s4
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 600 python -m pytest tests/test_streaming_gpu.py -q -x > $O/t_stream.log 2>&1
tail -n 25 $O/t_stream.log
timeout 1500 python -m pytest tests -q -x -m gpu --deselect tests/test_streaming_gpu.py > $O/t_all.log 2>&1
tail -n 12 $O/t_all.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
timeout 300 python bench.py $Q > $O/bench.log 2>&1; tail -n 1 $O/bench.log | cut -c1-1800
timeout 300 python bench_diar.py --steps 3 --warmup 1 --cpu-baseline-windows 0 > $O/bench_diar.log 2>&1; tail -n 1 $O/bench_diar.log | cut -c1-400
