#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s22
for pad in 0 16 64 256; do
echo "#### PADC $pad"
RVB_BENCH_PADC=$pad timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -v "over time\|gap between\|per K step\|prologue" | tee gpurun_out/s22/timeline_pad$pad.log
done
