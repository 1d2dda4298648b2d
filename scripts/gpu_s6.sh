#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s6
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_diar_gpu.py tests/test_diar_pipeline_gpu.py tests/test_fp8_gpu.py -q -x > $O/t_diar.log 2>&1
tail -n 6 $O/t_diar.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_streaming_gpu.py -q -x -k "rownorm or golden or conv1 or streaming or chunk" > $O/t_asr.log 2>&1
tail -n 6 $O/t_asr.log
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
timeout 300 python bench.py $Q > $O/bench.log 2>&1; tail -n 1 $O/bench.log | cut -c900-1700
timeout 300 python bench_diar.py --steps 3 --warmup 1 --cpu-baseline-windows 0 > $O/bench_diar.log 2>&1; tail -n 1 $O/bench_diar.log | cut -c1-300; tail -n 1 $O/bench_diar.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stage_ms_per_step'])"
