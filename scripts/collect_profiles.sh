#!/bin/bash
# After `gpurun -- bash scripts/refresh_profiles.sh`: copy the judged evidence from gpurun_out/refresh/ into profiles/ (tracked).
# gpurun MERGES a call's output into the local gpurun_out/, so kernel-stats files of earlier calls (named by PID) may still lie
# there: the newest one is taken.  Usage: scripts/collect_profiles.sh r02
set -eu
R=${1:?round tag, e.g. r02}
O=gpurun_out/refresh; P=profiles
A=$(ls -t $O/prof_asr/*/*kernel_stats.csv | head -1); D=$(ls -t $O/prof_diar/*/*kernel_stats.csv | head -1)
cp "$A" $P/${R}_rocprofv3_kernel_stats_r640_1h.csv
cp "$D" $P/${R}_rocprofv3_kernel_stats_diar_1h.csv
grep '^{' $O/bench_r640.log | tail -1 > $P/${R}_bench_r640_1h_bf16.json.log
grep '^{' $O/bench_r640_fp8.log | tail -1 > $P/${R}_bench_r640_1h_fp8.json.log
grep '^{' $O/bench_r268.log | tail -1 > $P/${R}_bench_r268_1h_bf16.json.log
grep '^{' $O/bench_diar.log | tail -1 > $P/${R}_bench_diar_1h_bf16.json.log
[ -f $O/gemm_traffic.json ] && cp $O/gemm_traffic.json $P/${R}_gemm_traffic_r640_1h_bf16.json || true
[ -f $O/pmc_by_kernel.csv ] && cp $O/pmc_by_kernel.csv $P/${R}_pmc_by_kernel.csv || true
cp $O/parity_metrics.jsonl $P/${R}_parity_metrics.jsonl
grep -a "passed" $O/pytest_gpu.log | tail -1 > $P/${R}_pytest_gpu_summary.txt
grep '^{' $O/bench_r640_forced_dist.log | tail -1 > $P/${R}_bench_r640_1h_forced_dist.json.log
[ -f $O/bench_r640_forced_dist_posteriors.log ] && grep '^{' $O/bench_r640_forced_dist_posteriors.log | tail -1 > $P/${R}_bench_r640_1h_forced_dist_posteriors.json.log || true
[ -f $O/bench_joint_3h.log ] && grep '^{' $O/bench_joint_3h.log | tail -1 > $P/${R}_bench_joint_3h_fp8.json.log || true
[ -f $O/gemm_bench.txt ] && cp $O/gemm_bench.txt $P/${R}_gemm_bench_switches.txt || true
[ -f $O/gemm_timeline.txt ] && cp $O/gemm_timeline.txt $P/${R}_gemm_timeline.txt || true
ls -la $P | grep " ${R}_" | awk '{print $5, $9}'
