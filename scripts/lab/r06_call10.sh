#!/bin/bash
# round 6, call 10: the hour as two (and four) half-hour engines on their own HIP streams against one engine
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call10; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
for L in 2 4 2; do timeout 400 python scripts/lab/two_engines.py --lanes $L 2>$O/err_$L.log | tee -a $O/lanes.txt; done
tail -n 3 $O/err_2.log
