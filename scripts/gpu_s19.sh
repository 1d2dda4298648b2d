#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s19
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > gpurun_out/s19/test_gemm.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s19/test_gemm.log
timeout 300 python scripts/gemm_bench.py 2>&1 | tail -14 | cut -c1-70
timeout 300 python bench.py --steps 3 --warmup 1 --traffic off --no-diarization --no-pcie --cpu-baseline-chunks 0 > gpurun_out/s19/bench.log 2>&1
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/s19/bench.log') if x.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['stage_ms_per_step'])
PY
