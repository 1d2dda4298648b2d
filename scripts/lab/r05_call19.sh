#!/bin/bash
# Round 5, GPU call 19: the sinc filter bank on the fp32 matrix pipe (RVD_SINC_MFMA=0: the VALU form) and its output stored in the
# engine's dtype (bf16 engine: half the bytes pool_norm reads twice per window).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call19; mkdir -p $O
timeout 900 python -m pytest tests/test_diar_gpu.py tests/test_diar_pipeline_gpu.py -q -m gpu -k "segmentation or short_audio or window or pipeline" 2>&1 | tail -8
for m in 0 1; do
  echo "== RVD_SINC_MFMA=$m"; RVB_LAB=1 RVD_SINC_MFMA=$m timeout 600 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "segmentation or short_audio" 2>&1 | tail -3
done
D="--steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"sinc_conv\": [0-9.]*\|\"pool_norm\": [0-9.]*" | tr "\n" " "; echo; }
run() { echo -n "diar $1: "; env RVB_LAB=1 $1 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_$2.json | pickd; }
for rep in 1 2; do
  run "RVD_X=0" mfma
  run "RVD_SINC_MFMA=0" valu
done
