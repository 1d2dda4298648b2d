#!/bin/bash
# Round 5, GPU call 1: settle what sat untested in HEAD + the new N > 1 test + two A/Bs that need no new code.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call1; mkdir -p $O
echo "== candidate tests + real-engine shard test"
RVB_TEST_CANDIDATES=1 timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_diar_gpu.py tests/test_shard_real_gpu.py -q -m gpu \
  -k "implicit_gemm_convolution_against or projection_shortcut or trunk_stages_3_and_4 or two_ranks_on_one_gpu" -rs 2>&1 | tee $O/tests.txt | tail -40
D="--steps 2 --warmup 1 --traffic off --cpu-baseline-windows 0"
pickd() { grep -o "\"ms_per_step\": [0-9.]*\|\"emb_conv_128\": [0-9.]*\|\"emb_conv_256\": [0-9.]*\|\"emb_conv_sc\": [0-9.]*" | tr "\n" " "; echo; }
for cfg in "0 0" "1 0" "0 1" "0 0" "1 0" "0 1"; do
  set -- $cfg
  echo -n "diar RVD_CONV_SC_FUSE=$1 RVD_EMB_FP8=$2: "
  RVD_CONV_SC_FUSE=$1 RVD_EMB_FP8=$2 timeout 200 python bench_diar.py $D 2>/dev/null | tee $O/diar_sc$1_f8$2.json | pickd
done
echo "== GEMM traffic A/B (bench.py, nested PMC passes)"
B="--steps 5 --warmup 1 --no-diarization --no-pcie --cpu-baseline-chunks 0"
pickb() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; t = r.get('traffic_detail') or {}
print('ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'avg_us', r['avg_launch_us'], 'read', t.get('read_bytes_per_launch'), 'write', t.get('write_bytes_per_launch'), 'stages', d.get('stage_ms_per_step'))
"; }
for cfg in "default:" "stagger0:RVB_GEMM2_STAGGER=0" "flags1024:RVB_GEMM2_FLAGS=1024" "stagger0_flags1024:RVB_GEMM2_STAGGER=0 RVB_GEMM2_FLAGS=1024"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo -n "bench $name: "
  env $envs timeout 300 python bench.py $B 2>/dev/null | tee $O/bench_$name.json | pickb
done
