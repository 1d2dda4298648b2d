#!/bin/bash
# Round 5, GPU call 22 (final tree): per-kernel HBM traffic and MFMA busy of one step of both workloads (PMC passes + a kernel trace
# for the durations), as scripts/lab/r05_call13.sh did before the band / strip / stride-2 / sinc kernels.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r05_call22; mkdir -p $O
N="--steps 1 --warmup 1 --no-diarization --no-pcie --no-variants --traffic off --cpu-baseline-chunks 0 --no-profile"
D="--steps 1 --warmup 1 --traffic off --cpu-baseline-windows 0"
for w in asr diar; do
  if [ $w = asr ]; then CMD="python $R/bench.py $N"; else CMD="python $R/bench_diar.py $D"; fi
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/${w}_pmc/p$i -- $CMD > $O/${w}_p$i.log 2>&1 < /dev/null
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_trace -- $CMD > $O/${w}_trace.log 2>&1 < /dev/null
  S=$(ls -t $O/${w}_trace/*/*kernel_stats.csv | head -1)
  echo "== $w (one warm-up + one timed step: launches = 2 steps; diar: + 2 pcie_inclusive steps = 4)"
  python $R/scripts/pmc_table.py $O/${w}_pmc "$S" | tee $O/${w}_pmc_table.txt
  rm -rf $O/${w}_pmc $O/${w}_trace
done
