#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s21
for st in 0 2 4 8; do
RVB_GEMM2_STAGGER=$st timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -v "over time\|gap between\|per K step" | tee gpurun_out/s21/timeline_$st.log
done
for st in 0 4; do
RVB_GEMM2_STAGGER=$st timeout 300 python scripts/gemm_bench.py 2>&1 | tail -13 | cut -c1-60
done
