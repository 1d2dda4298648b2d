#!/bin/bash
# First GPU call of round 5: the candidates round 4 left compiled but never ran (its GPU minutes were spent).
#   1. RVD_CONV_SC_FUSE=1   projection shortcut inside the second convolution's K loop (conv_gemm.hip, ConvArgs::in2)
#   2. conv_igemm8_kernel   fp8 implicit-GEMM convolution, kernel-level test through rvb_test_conv_igemm_fp8
#   3. RVD_EMB_FP8=1        stages 3-4 of the ResNet34 trunk on that kernel (calibration pass, act_quant_fp8, rvd_get_emb_fp8)
# Each has a GPU test gated behind RVB_TEST_CANDIDATES=1; the A/B legs below only make sense for what passed.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_candidates; mkdir -p $O
echo "== candidate tests"
RVB_TEST_CANDIDATES=1 timeout 300 python -m pytest tests/test_fp8_gpu.py tests/test_diar_gpu.py -q -m gpu -k "implicit_gemm_convolution_against or projection_shortcut or trunk_stages_3_and_4" 2>&1 | tail -15
D="--steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*\|\"emb_conv_sc\": [0-9.]*" | tr "\n" " "; echo; }
for cfg in "0 0" "1 0" "0 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  echo -n "diar RVD_CONV_SC_FUSE=$1 RVD_EMB_FP8=$2: "
  RVD_CONV_SC_FUSE=$1 RVD_EMB_FP8=$2 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_sc$1_f8$2.json | pickd
done
