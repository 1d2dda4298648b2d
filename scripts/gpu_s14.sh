#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s14
timeout 600 python -m pytest tests/test_diar_gpu.py -q -x -k "linkage" > gpurun_out/s14/test_linkage.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/s14/test_linkage.log
RVD_LINKAGE_PROF=1 timeout 300 python scripts/linkage_bench.py 2>&1 | tee gpurun_out/s14/linkage_prof.log
timeout 300 python scripts/linkage_bench.py 2>&1 | tee gpurun_out/s14/linkage.log
timeout 600 python bench_diar.py --steps 2 --warmup 1 --cpu-baseline-windows 0 > gpurun_out/s14/bench_diar.log 2>&1; tail -1 gpurun_out/s14/bench_diar.log | cut -c1-300; grep -o '"linkage": [0-9.]*' gpurun_out/s14/bench_diar.log
