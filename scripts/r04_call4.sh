#!/bin/bash
# Round 4, GPU call 4: the multi-workgroup linkage loop with lazy validation rounds; librvb's communicator as the only RCCL
# communicator of a rank (barrier, max, device-to-device posterior gather, timeout); short-case bf16 bounds.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call4; mkdir -p $O
echo "== linkage + comm + bf16-bound tests"
timeout 400 python -m pytest tests/test_diar_gpu.py tests/test_edge_cases_gpu.py tests/test_engine_gpu.py tests/test_causal_gpu.py -q -m gpu -x -k "linkage or collective or bf16 or double_buffered" 2>&1 | grep -v "^shader\|^linkage n=" | tail -12
echo "== linkage bench (one workgroup vs sixteen)"
for mb in 0 1; do
  echo "RVD_LINKAGE_MB=$mb"; RVD_LINKAGE_MB=$mb timeout 120 python scripts/linkage_bench.py 2>&1 | tail -3
done
echo "== n = 27 000 (three hours)"
for mb in 0 1; do
  echo "RVD_LINKAGE_MB=$mb"; RVD_LINKAGE_MB=$mb timeout 200 python scripts/linkage_bench.py 27000 2>&1 | tail -1
done
echo "== diarization, 1 h"
timeout 200 python bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar.json | grep -o "\"ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"linkage[a-z_]*\": [0-9.]*" | tr "\n" " "; echo
echo "== forced distributed path on one rank (one RCCL communicator; posteriors device to device)"
B="--steps 3 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
RVB_FORCE_DIST=1 timeout 200 python bench.py $B 2>$O/dist.err | tee $O/bench_forced_dist.json | grep -o "\"ms_per_step\": [0-9.]*\|\"backend\": \"[^\"]*\"\|\"xgmi_allgather\": {[^}]*}" | tr "\n" " "; echo
RVB_FORCE_DIST=1 timeout 200 python bench.py $B --gather posteriors 2>>$O/dist.err | tee $O/bench_forced_dist_posteriors.json | grep -o "\"ms_per_step\": [0-9.]*\|\"posterior_bytes_gathered_per_step\": [0-9]*" | tr "\n" " "; echo
RVB_FORCE_DIST=1 timeout 200 python bench_diar.py --steps 1 --warmup 1 --traffic off --cpu-baseline-windows 0 2>>$O/dist.err | grep -o "\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo
tail -5 $O/dist.err
echo "== joint, 3 h"
timeout 400 python bench_joint.py --hours 3 --steps 1 --warmup 1 2>$O/joint3h.err | tee $O/joint_3h.json | cut -c1-1200
