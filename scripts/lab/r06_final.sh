#!/bin/bash
# Round 6 evidence run (one GPU call): scripts/refresh_profiles.sh (GPU suite, smoke, the driver's bench command, bench_diar stand-alone,
# the collective path on one rank, the vendor GEMM yardstick, MP3 decode speed, kernel traces of both workloads) + the per-kernel PMC
# table of one step of both workloads (as scripts/lab/r05_call22.sh).  scripts/collect_profiles.sh <tag> copies the results into profiles/.
set -u
R=$GRAFT_REPO_ROOT
bash $R/scripts/refresh_profiles.sh
export PYTHONPATH=$R
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/refresh
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
  echo "== $w (one warm-up + one timed step: launches = 2 steps; diar: + 2 pcie_inclusive steps = 4)" | tee -a $O/pmc_by_kernel.txt
  python $R/scripts/pmc_table.py $O/${w}_pmc "$S" | tee -a $O/pmc_by_kernel.txt
  rm -rf $O/${w}_pmc $O/${w}_trace
done
ls $O
