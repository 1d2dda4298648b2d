#!/bin/bash
# Round 5, GPU call 7: counters of conv_block32_kernel (call 6 lost them to a name filter) and of the two-launch form.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05_call7; mkdir -p $O
D="--steps 1 --warmup 1 --traffic off --cpu-baseline-windows 0"
run() {
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python $R/bench_diar.py $D > $O/$name.log 2>&1 < /dev/null
  echo "== $name ($envs): $*"
  python $R/scripts/pmc_by_kernel.py $O/$name conv_block conv_stream | tee $O/$name.txt | head -4
  rm -rf $O/$name
}
run fused_fetch -- FETCH_SIZE
run fused_write -- WRITE_SIZE
run fused_lds -- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run fused_sq -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run fused_mfma -- SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS
run two_fetch RVB_LAB=1 RVD_CONV_BLOCK=0 -- FETCH_SIZE
run two_write RVB_LAB=1 RVD_CONV_BLOCK=0 -- WRITE_SIZE
run two_lds RVB_LAB=1 RVD_CONV_BLOCK=0 -- SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
