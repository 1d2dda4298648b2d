#!/bin/bash
# Round 4, GPU call 13: why the full-tile epilogue produced a few wrong elements in gemm2_kernel's 32x32 path with a residual.
# Three builds of gemm2.hip: v1 = hidden stores whose data registers stay allocated for one more slab (the tree's build),
# v0 = hidden stores, registers free at once (call 11's build), v2 = plain stores the compiler sees (ring unchanged).
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
cp reverb_amd/librvb.so /tmp/librvb_v1.so
for v in 1 0 2 1 0; do
  if [ $v = 1 ]; then cp /tmp/librvb_v1.so reverb_amd/librvb.so; else cp gpurun_in_librvb_v$v.so reverb_amd/librvb.so; fi
  echo -n "build v$v: "
  timeout 120 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "tuning_switches or gemm2" 2>&1 | grep -a -E "passed|failed" | tail -1
done
cp /tmp/librvb_v1.so reverb_amd/librvb.so
echo "== gemm_bench v1 (hold) fast vs off"
timeout 200 python scripts/gemm_bench.py 0,-2 1024,-2 2>&1 | tail -13
