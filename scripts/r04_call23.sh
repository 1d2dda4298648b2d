#!/bin/bash
# Round 4, GPU call 23: conv2 of the subsampling in fp8 (policy bit 5): unit test, token error rates on the bench hour, speed.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_fp8_gpu.py -q -m gpu -k "conv2_policy or policy_is_validated or saturation" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -6
echo "== TER on the bench hour: groups 17 (default) vs 49 (+ conv2)"
timeout 400 python scripts/fp8_sweep.py 17:0:17 49:0:17 2>&1 | tail -3
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"gemm_fp8\": [0-9.]*\|\"subsample\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for g in 17 49 17 49; do
  echo -n "fp8 RVB_FP8_GROUPS=$g: "
  RVB_FP8_GROUPS=$g timeout 150 python bench.py --dtype fp8 $B 2>/dev/null | pick
done
