#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s25
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -q -x -k "gemm" > gpurun_out/s25/test_gemm.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s25/test_gemm.log
timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -v "gap between\|per K step\|over time" | tee gpurun_out/s25/timeline.log
timeout 300 python scripts/gemm_bench.py 2>&1 | tail -13 | cut -c1-60
timeout 300 python bench.py --steps 3 --warmup 1 --traffic off --no-diarization --no-pcie --cpu-baseline-chunks 0 > gpurun_out/s25/bench.log 2>&1
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/s25/bench.log') if x.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['stage_ms_per_step'])
PY
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_longform_gpu.py -q -x > gpurun_out/s25/test_engine.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/s25/test_engine.log
