#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s27
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > gpurun_out/s27/test_gemm.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/s27/test_gemm.log
timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -A3 "out/pw2\|ffn2" | grep -v "over time\|per K" | tee gpurun_out/s27/timeline.log
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --traffic off --no-diarization --no-pcie --cpu-baseline-chunks 0 > gpurun_out/s27/bench_$i.log 2>&1
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/s27/bench_$i.log') if x.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['stage_ms_per_step']['gemm'])
PY
done
