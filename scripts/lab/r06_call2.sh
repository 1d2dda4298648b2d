#!/bin/bash
# round 6, call 2: (a) gemm2p on 32x32x16 MFMAs (flag bit 13) against the default phase loop, kernel benchmark on the engine's shapes;
# (b) the GPU suite at the tree of the first commit of the round; (c) the driver's bench line (compact form, PMC over one step);
# (d) launches per step: kernel-trace of 2 and of 6 timed steps, differenced.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call2; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp RVB_LAB=1
cd $R
timeout 600 python scripts/gemm_bench.py 0,-2 8192,-2 0,-2 8192,-2 > $O/gemm_bench_m32.txt 2>&1; tail -n 14 $O/gemm_bench_m32.txt
unset RVB_LAB
timeout 1500 python -m pytest tests -q -m gpu -x -rs > $O/pytest_gpu.log 2>&1; tail -n 5 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r640.log 2> $O/bench_r640.err; tail -c 1800 $O/bench_r640.log
cp gpurun_out/bench_long.json $O/ 2>/dev/null
cd /tmp
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0 --no-profile"
for K in 2 6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$K -- python $R/bench.py --steps $K --warmup 0 $N > $O/trace${K}_stdout.log 2>&1 < /dev/null
  f=$(find $O/trace$K -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_steps$K.csv
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace2 $O/trace6
ls $O
