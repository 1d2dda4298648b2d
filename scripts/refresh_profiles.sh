#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): the GPU test suite, smoke, the driver's bench command (its compact line + the long
# form in bench_long.json carry
# the sub-records: diarization, joint_fp8 on 3 h, parity_f32, asr_fp8, r268, pcie_inclusive, PMC traffic), bench_diar stand-alone,
# the collective path on one rank, rocprofv3 kernel stats of both workloads, and the vendor GEMM yardstick.
# Everything lands under gpurun_out/refresh/ ; scripts/collect_profiles.sh <tag> copies what is to be judged into profiles/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout ${REFRESH_PYTEST_TIMEOUT:-1200} python -m pytest tests -q -m gpu -rs > $O/pytest_gpu.log 2>&1
tail -n 6 $O/pytest_gpu.log
cp $R/gpurun_out/parity_metrics.jsonl $O/parity_metrics.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r640.log 2> $O/bench_r640.err
cp $R/gpurun_out/bench_long.json $O/bench_long.json 2>/dev/null
timeout 400 python bench_diar.py --steps 3 --warmup 1 > $O/bench_diar.log 2>&1
RVB_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0 > $O/bench_r640_forced_dist.log 2>&1
timeout 300 python scripts/blaslt_ref.py > $O/vendor_gemm_yardstick.txt 2>&1
timeout 300 python scripts/mp3_bench.py > $O/mp3_decode_speed.txt 2>&1
cd /tmp
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_asr -- python $R/bench.py --steps 2 --warmup 1 $N > $O/prof_asr_stdout.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_diar -- python $R/bench_diar.py --steps 2 --warmup 1 --cpu-baseline-windows 0 --traffic off > $O/prof_diar_stdout.log 2>&1 < /dev/null
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls $O | head -40
tail -n 1 $O/bench_r640.log | cut -c1-400
