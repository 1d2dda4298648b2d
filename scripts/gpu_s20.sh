#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s20
timeout 300 python scripts/gemm_timeline.py 2>&1 | tee gpurun_out/s20/timeline.log
