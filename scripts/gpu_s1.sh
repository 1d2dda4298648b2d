#!/bin/bash
# GPU session 1 (round 2): new parity tests, gemm2 switch sweep, bench with the new sub-records, conv_igemm first run
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s1
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/t_gemm.log 2>&1
tail -n 3 $O/t_gemm.log
timeout 300 python scripts/gemm_bench.py > $O/gemm_bench.log 2>&1
cat $O/gemm_bench.log
timeout 1200 python -m pytest tests/test_longform_gpu.py tests/test_engine_gpu.py -q -k "longform or small_66 or r640 or chunk_size" > $O/t_long.log 2>&1
tail -n 15 $O/t_long.log
cp $R/gpurun_out/parity_metrics.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err
tail -n 1 $O/bench.log | cut -c1-3000
tail -n 5 $O/bench.err
RVD_CONV_IGEMM=1 timeout 600 python -m pytest tests/test_diar_gpu.py -q -x > $O/t_diar_igemm.log 2>&1
tail -n 3 $O/t_diar_igemm.log
RVD_CONV_IGEMM=1 timeout 300 python bench_diar.py --steps 3 --warmup 1 --cpu-baseline-windows 0 > $O/bench_diar_igemm.log 2>&1
tail -n 1 $O/bench_diar_igemm.log | cut -c1-1500
