#!/bin/bash
# session 11: linkage with register-cached minima
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s11
timeout 600 python -m pytest tests/test_diar_gpu.py -q -x -k "linkage" > gpurun_out/s11/test_linkage.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/s11/test_linkage.log
timeout 300 python scripts/linkage_bench.py > gpurun_out/s11/linkage.log 2>&1; cat gpurun_out/s11/linkage.log
RVD_LINKAGE_PROF=1 timeout 300 python scripts/linkage_bench.py > gpurun_out/s11/linkage_prof.log 2>&1; cat gpurun_out/s11/linkage_prof.log
RVD_LINKAGE_NOCOL=1 timeout 300 python scripts/linkage_bench.py > gpurun_out/s11/linkage_nocol.log 2>&1; cat gpurun_out/s11/linkage_nocol.log
timeout 600 python bench_diar.py --steps 2 --warmup 1 --cpu-baseline-windows 0 > gpurun_out/s11/bench_diar.log 2>&1; tail -1 gpurun_out/s11/bench_diar.log | cut -c1-600
