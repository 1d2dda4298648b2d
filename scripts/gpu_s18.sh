#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s18
timeout 300 python scripts/blaslt_ref.py 2>&1 | tee gpurun_out/s18/blaslt.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s18/prof -- python $GRAFT_REPO_ROOT/scripts/blaslt_ref.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/s18/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:150], r['Calls'], r['AverageNs'])
PY
timeout 300 python scripts/gemm_bench.py 2>&1 | tail -30
