#!/bin/bash
# round 6, call 6: the tests added or touched after the evidence call (2-D cat_embs, the M32 lab variant trimmed to the plain shapes,
# the mp3 end-to-end test) and the whole GPU suite once more at the final tree.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call6; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x -rs > $O/pytest_gpu.log 2>&1; tail -n 6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
ls $O
