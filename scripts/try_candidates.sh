#!/bin/bash
# One gpurun call that settles the candidates of this branch (CANDIDATES.md): unit tests of the two rewritten kernels, the engine's
# parity tests, then the step time with and without the GEMM start stagger.  ~4 GPU-minutes.
#   gpurun --timeout 600 -- 'bash scripts/try_candidates.sh'
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -q -m gpu -k "rownorm or glu or dwconv or norm or layernorm_to_fp8" 2>&1 | grep -a -E "passed|failed" | tail -2
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_causal_gpu.py tests/test_streaming_gpu.py tests/test_longform_gpu.py -q -m gpu -x 2>&1 | grep -a -E "passed|failed" | tail -2
B="--steps 3 --warmup 1 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0"
for f in 0 64 0 64; do
  echo -n "RVB_GEMM2_FLAGS=$f "
  RVB_GEMM2_FLAGS=$f timeout 100 python bench.py $B 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"rownorm\": [0-9.]*\|\"glu_dwconv\": [0-9.]*" | tr "\n" " "; echo
done
# main for comparison on the same box: git stash / checkout is not possible on the GPU box (no .git), so compare with the numbers
# of main from the same box class (attention is the unchanged gauge: 9.9 ms <-> GEMM 101.1 ms, norms 11.0, GLU 4.2 at the end of round 3)
