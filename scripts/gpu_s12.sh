#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s12
RVD_LINKAGE_PROF=1 timeout 300 python scripts/linkage_bench.py > gpurun_out/s12/linkage_prof.log 2>&1; cat gpurun_out/s12/linkage_prof.log
(rocm-smi --showclocks; rocm-smi --showperflevel) 2>&1 | head -40
