#!/bin/bash
# Round 5, GPU call 6: counters for the 32-channel stage -- fused block (conv_block32_kernel) vs two streamed launches.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05_call6; mkdir -p $O
D="--steps 1 --warmup 1 --traffic off --cpu-baseline-windows 0"
run() {  # name, env..., -- counters
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python $R/bench_diar.py $D > $O/$name.log 2>&1 < /dev/null
  echo "== $name ($envs): $*"
  python $R/scripts/pmc_by_kernel.py $O/$name conv_block conv_stream conv_kernel | head -8
}
run fused_fetch -- FETCH_SIZE
run fused_write -- WRITE_SIZE
run fused_lds -- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run fused_sq -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run fused_mfma -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run two_fetch RVB_LAB=1 RVD_CONV_BLOCK=0 -- FETCH_SIZE
run two_write RVB_LAB=1 RVD_CONV_BLOCK=0 -- WRITE_SIZE
run two_lds RVB_LAB=1 RVD_CONV_BLOCK=0 -- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
echo "== kernel trace (default)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench_diar.py --steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0 > $O/trace.log 2>&1 < /dev/null
S=$(ls -t $O/trace/*/*kernel_stats.csv | head -1); head -25 "$S" | cut -c1-200
cp "$S" $O/kernel_stats_diar.csv
rm -rf $O/*/  # the raw counter dumps are large
