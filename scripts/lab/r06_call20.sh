#!/bin/bash
# round 6, call 20: where the device idles inside the headline step (scripts/gap_report.py on a kernel trace of 4 timed steps)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call20; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp
N="--no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0 --no-profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --steps 4 --warmup 2 $N > $O/stdout.log 2>&1 < /dev/null
T=$(ls $O/tr/*/*kernel_trace.csv | head -1)
# window: from the 3rd fbank dispatch (first timed step) on
python $R/scripts/gap_report.py "$T" 20 fbank_kernel 2 | tee $O/gaps.txt
rm -rf $O/tr
tail -1 $O/stdout.log | cut -c1-300
