#!/bin/bash
# Round 4, GPU call 7: attention K / P row pitch 160 B (conflict-free for ds_read_b128's real lane groups) vs 144 B; LDS-conflict
# and wait counters of the attention kernel.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04_call7; mkdir -p $O
echo "== kernel tests"
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "attention or bf16 or f32_engine" 2>&1 | tail -3
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0 --no-pcie"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 16 32 16 32; do
  echo -n "RVB_ATTN_PADK=$f: "
  RVB_ATTN_PADK=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_padk$f.json | pick
done
cd /tmp
N="--steps 1 --warmup 0 --hours 0.25 --no-diarization --no-pcie --traffic off --cpu-baseline-chunks 0 --no-profile"
for pk in 16 32; do
  RVB_ATTN_PADK=$pk timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_lds_$pk -- python $R/bench.py $N > /dev/null 2>&1 < /dev/null
  RVB_ATTN_PADK=$pk timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_wait_$pk -- python $R/bench.py $N > /dev/null 2>&1 < /dev/null
  echo "-- PADK=$pk"
  python $R/scripts/pmc_by_kernel.py $O/pmc_lds_$pk attn_kernel gemm2p glu_dw rownorm | head -8
  python $R/scripts/pmc_by_kernel.py $O/pmc_wait_$pk attn_kernel gemm2p | head -6
done
RVB_ATTN_PADK=32 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_valu -- python $R/bench.py $N > /dev/null 2>&1 < /dev/null
python $R/scripts/pmc_by_kernel.py $O/pmc_valu attn_kernel gemm2p | head -6
find $O -name "*counter_collection.csv" -delete
