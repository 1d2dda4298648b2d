#!/bin/bash
# Round 5, GPU call 20: new tests (conv_s2 / conv_block ragged hooks, sinc forms, rerun_resident) and bench_diar with the samples
# resident in HBM at the start of the timed region (pcie_inclusive beside it).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call20; mkdir -p $O
timeout 900 python -m pytest tests/test_diar_gpu.py -q -m gpu -k "ragged or sinc or resident or stride2" 2>&1 | tail -8
for rep in 1 2; do
  timeout 300 python bench_diar.py --steps 3 --warmup 1 --traffic off --cpu-baseline-windows 0 2>/dev/null | tee $O/diar_$rep.json | grep -o "\"ms_per_step\": [0-9.]*\|\"pcie_inclusive\": {[^}]*}" | tr "\n" " "; echo
done
