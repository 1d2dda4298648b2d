#!/bin/bash
# After `gpurun -- bash scripts/refresh_profiles.sh`: copy the judged evidence from gpurun_out/refresh/ into profiles/ (tracked).
# Usage: scripts/collect_profiles.sh r05d
set -eu
R=${1:?round tag, e.g. r05d}
O=gpurun_out/refresh; P=profiles
A=$(ls -t $O/prof_asr/*/*kernel_stats.csv | head -1); D=$(ls -t $O/prof_diar/*/*kernel_stats.csv | head -1)
cp "$A" $P/${R}_rocprofv3_kernel_stats_r640_1h.csv
cp "$D" $P/${R}_rocprofv3_kernel_stats_diar_1h.csv
grep '^{' $O/bench_r640.log | tail -1 > $P/${R}_bench_r640_1h_bf16.json.log
[ -s $O/bench_long.json ] && cp $O/bench_long.json $P/${R}_bench_long.json || true
[ -s $O/mp3_decode_speed.txt ] && cp $O/mp3_decode_speed.txt $P/${R}_mp3_decode_speed.txt || true
grep '^{' $O/bench_diar.log | tail -1 > $P/${R}_bench_diar_1h_bf16.json.log
grep '^{' $O/bench_r640_forced_dist.log | tail -1 > $P/${R}_bench_r640_1h_forced_dist.json.log
cp $O/parity_metrics.jsonl $P/${R}_parity_metrics.jsonl
(grep -a "passed\|SKIPPED\|failed" $O/pytest_gpu.log | tail -8; tail -3 $O/smoke.log) > $P/${R}_pytest_gpu_summary.txt
[ -s $O/vendor_gemm_yardstick.txt ] && cp $O/vendor_gemm_yardstick.txt $P/${R}_vendor_gemm_yardstick.txt || true
[ -s $O/pmc_by_kernel.txt ] && cp $O/pmc_by_kernel.txt $P/${R}_pmc_by_kernel.txt || true
ls -la $P | grep " ${R}_" | awk '{print $5, $9}'
