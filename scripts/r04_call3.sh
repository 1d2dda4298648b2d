#!/bin/bash
# Round 4, GPU call 3: the multi-workgroup linkage loop (tests against scipy and the one-workgroup loop, 9 200- and 27 000-point
# timings), the per-shape K serpentine default, diarization and joint-pipeline steps.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call3; mkdir -p $O
echo "== linkage tests"
timeout 300 python -m pytest tests/test_diar_gpu.py -q -m gpu -x -k "linkage" 2>&1 | tail -15
echo "== linkage bench (one workgroup vs sixteen)"
for mb in 0 1; do
  echo "RVD_LINKAGE_MB=$mb"; RVD_LINKAGE_MB=$mb timeout 120 python scripts/linkage_bench.py 2>&1 | tail -4
done
echo "== n = 27 000 (three hours)"
for mb in 0 1; do
  echo "RVD_LINKAGE_MB=$mb"; RVD_LINKAGE_MB=$mb timeout 200 python scripts/linkage_bench.py 27000 2>&1 | tail -2
done
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"rownorm\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 256 0 256 0; do
  echo -n "RVB_GEMM2_FLAGS=$f: "
  RVB_GEMM2_FLAGS=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_f$f.json | pick
done
echo "== diarization, 1 h"
timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar.json | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
echo "== joint, 3 h"
timeout 400 python bench_joint.py --hours 3 --steps 1 --warmup 1 2>$O/joint3h.err | tee $O/joint_3h.json | cut -c1-1200
