#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): bench logs, rocprofv3 kernel stats, PMC traffic of the GEMMs.
# Everything is written under gpurun_out/refresh/ ; copy what is to be judged into profiles/ afterwards.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp
# GEMM main-loop laboratory (build it first, here or on the box: hipcc -O3 --offload-arch=gfx950 -o scripts/micro/gemm_lab scripts/micro/gemm_lab.hip)
[ -x $R/scripts/micro/gemm_lab ] && timeout 120 $R/scripts/micro/gemm_lab 5 > $O/gemm_lab.txt 2>&1 < /dev/null
timeout 300 python $R/bench.py --steps 3 --warmup 1 > $O/bench_r640.log 2>&1 < /dev/null
timeout 300 python $R/bench.py --steps 3 --warmup 1 --model r268 > $O/bench_r268.log 2>&1 < /dev/null
timeout 300 python $R/bench_diar.py --steps 3 --warmup 1 > $O/bench_diar.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_asr -- python $R/bench.py --steps 2 --warmup 1 --cpu-baseline-chunks 0 > $O/prof_asr_stdout.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_diar -- python $R/bench_diar.py --steps 2 --warmup 1 --cpu-baseline-windows 0 > $O/prof_diar_stdout.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline-chunks 0 > $O/pmc_fetch_stdout.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --cpu-baseline-chunks 0 > $O/pmc_write_stdout.log 2>&1 < /dev/null
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -n 1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -n 1)
if [ -n "$F" ] && [ -n "$W" ]; then python scripts/pmc_traffic.py "$F" "$W" $O/gemm_traffic.json; fi
# keep the merged output small: the per-dispatch traces are big
find $O -name "*kernel_trace.csv" -delete
ls -la $O $O/* | head -60
tail -n 1 $O/bench_r640.log | cut -c1-400
