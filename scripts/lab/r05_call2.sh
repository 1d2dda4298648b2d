#!/bin/bash
# Round 5, GPU call 2: the whole GPU suite after the product / lab split, smoke, and the driver's bench command with the new sub-records.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call2; mkdir -p $O
rm -f gpurun_out/parity_metrics.jsonl
echo "== pytest -m gpu"
timeout 900 python -m pytest tests/ -q -m gpu -x -rs 2>&1 | tee $O/pytest_gpu.txt | tail -25
cp gpurun_out/parity_metrics.jsonl $O/parity_metrics.jsonl 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== bench (driver's command)"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -4
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_call2/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "cpu", d["cpu_baseline"])
for k in ("parity_f32", "asr_fp8", "r268"):
    print(k, d.get(k))
j = d.get("joint_fp8", {})
print("joint", {k: j.get(k) for k in ("ms_per_step", "sequential_ms_per_step", "sharded_s", "replicated_s", "sharded_parts_s", "replicated_parts_s", "projected_8gpu_step_s", "diarization_fp8", "error", "last_step_s")})
dd = d.get("diarization", {})
print("diar", dd.get("ms_per_step"), dd.get("roofline"), dd.get("stage_ms_per_step"))
PY
