#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s7
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_diar_gpu.py tests/test_fp8_gpu.py -q -x -k "linkage or gemm" > $O/t.log 2>&1; tail -n 3 $O/t.log
Q="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
for qb in 0 64; do RVB_ATTN_QBLOCK=$qb timeout 300 python bench.py $Q > $O/bench_qb$qb.log 2>&1; tail -n 1 $O/bench_qb$qb.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qblock $qb', d['ms_per_step'], d['stage_ms_per_step']['attention'])"; done
timeout 300 python bench_diar.py --steps 3 --warmup 1 --cpu-baseline-windows 0 > $O/bench_diar.log 2>&1; tail -n 1 $O/bench_diar.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step']['linkage'])"
