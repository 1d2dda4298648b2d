#!/bin/bash
# Round 5, GPU call 21: conv1 (CMVN + Conv2d(1, d, 3, 2) + ReLU of the subsampling front end) on packed FMAs in the bf16 / fp8 engines;
# the new diarization tests; the ASR step's stage table.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call21; mkdir -p $O
timeout 900 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "ragged or sinc or resident or stride2" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu -k "conv1 or subsample or golden or bf16" 2>&1 | tail -5
for rep in 1 2; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0 2>/dev/null | tee $O/asr_$rep.json | grep -o "\"ms_per_step\": [0-9.]*\|\"subsample\": [0-9.]*\|\"gemm\": [0-9.]*" | tr "\n" " "; echo
done
