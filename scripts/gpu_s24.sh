#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s24
timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -v "gap between\|per K step\|prologue" | tee gpurun_out/s24/timeline.log
RVB_GEMM2_STAGGER=8 timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -A4 "out/pw2" | tee gpurun_out/s24/timeline_st8.log
