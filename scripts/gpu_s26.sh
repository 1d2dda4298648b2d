#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=.
mkdir -p gpurun_out/s26
for st in 3 4 8; do
RVB_GEMM2_STAGGER=$st timeout 300 python scripts/gemm_timeline.py 2>&1 | grep -A3 "STAGGER\|out/pw2\|ffn2\|ffn1" | grep -v "over time\|per K" | tee gpurun_out/s26/timeline_$st.log
RVB_GEMM2_STAGGER=$st timeout 300 python bench.py --steps 3 --warmup 1 --traffic off --no-diarization --no-pcie --cpu-baseline-chunks 0 > gpurun_out/s26/bench_$st.log 2>&1
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/s26/bench_$st.log') if x.startswith('{')][-1])
print("stagger", $st, d['ms_per_step'], d['value'], d['roofline']['achieved'], d['stage_ms_per_step']['gemm'])
PY
done
