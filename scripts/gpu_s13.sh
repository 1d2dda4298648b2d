#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s13
timeout 600 python -m pytest tests/test_diar_gpu.py -q -x -k "linkage" > gpurun_out/s13/test_linkage.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/s13/test_linkage.log
for th in 1024 512 256; do
echo "== threads $th"
RVD_LINKAGE_THREADS=$th RVD_LINKAGE_PROF=1 timeout 300 python scripts/linkage_bench.py 2>&1 | tee gpurun_out/s13/linkage_prof_$th.log
done
RVD_LINKAGE_THREADS=512 timeout 600 python -m pytest tests/test_diar_gpu.py -q -x -k "linkage" 2>&1 | tail -2
