#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): GPU test suite, bench lines of every workload, rocprofv3 kernel
# stats, PMC traffic of the GEMMs.  Everything lands under gpurun_out/refresh/ ; copy what is to be judged into profiles/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_r640.log 2> $O/bench_r640.err
timeout 300 python bench.py --steps 3 --warmup 1 --dtype fp8 --no-diarization --no-pcie --cpu-baseline-chunks 0 > $O/bench_r640_fp8.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --model r268 --no-diarization --no-pcie --traffic off > $O/bench_r268.log 2>&1
timeout 400 python bench_diar.py --steps 3 --warmup 1 > $O/bench_diar.log 2>&1
RVB_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0 > $O/bench_r640_forced_dist.log 2>&1
RVB_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0 --gather posteriors > $O/bench_r640_forced_dist_posteriors.log 2>&1
timeout 400 python bench_joint.py --hours 3 --steps 2 --warmup 1 > $O/bench_joint_3h.log 2>&1      # config 5's audio length on one GPU
if [ "${REFRESH_GEMM_LAB:-0}" = 1 ]; then      # the GEMM micro-benchmarks only change when gemm2.hip does
  timeout 200 python scripts/gemm_bench.py 0,-2 4,8 > $O/gemm_bench.txt 2>&1
  timeout 200 python scripts/gemm_timeline.py > $O/gemm_timeline.txt 2>&1
fi
cd /tmp
N="--no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_asr -- python $R/bench.py --steps 2 --warmup 1 $N > $O/prof_asr_stdout.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_diar -- python $R/bench_diar.py --steps 2 --warmup 1 --cpu-baseline-windows 0 --traffic off > $O/prof_diar_stdout.log 2>&1 < /dev/null
if [ "${REFRESH_PMC:-0}" = 1 ]; then            # bench.py's own nested passes already put the GEMM traffic into the bench line
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 $N --no-profile > $O/pmc_fetch_stdout.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 $N --no-profile > $O/pmc_write_stdout.log 2>&1 < /dev/null
fi
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" 2>/dev/null | head -n 1)
W=$(find $O/pmc_write -name "*counter_collection.csv" 2>/dev/null | head -n 1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python scripts/pmc_traffic.py "$F" "$W" $O/gemm_traffic.json
  # per-kernel sums (the per-dispatch files are tens of MB)
  python - "$F" "$W" > $O/pmc_by_kernel.csv <<'PY'
import csv, sys, collections
for path, key in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == key:
            k = r["Kernel_Name"].split("(")[0][:90]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k, v in tot.most_common(14):
        print(f'{key},"{k}",{n[k]},{v * 1024:.0f}')
PY
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# the GPU test suite last: it is the longest leg, and the driver runs it again at round end
rm -f $R/gpurun_out/parity_metrics.jsonl
timeout ${REFRESH_PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 5 $O/pytest_gpu.log
cp $R/gpurun_out/parity_metrics.jsonl $O/parity_metrics.jsonl 2>/dev/null
ls $O | head -40
tail -n 1 $O/bench_r640.log | cut -c1-600
