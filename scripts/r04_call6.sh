#!/bin/bash
# Round 4, GPU call 6: the positional term of the encoder attention folded into per-key constants (bf16): kernel tests, the bench
# hour's token error rates against the reference-bf16 yardstick, and the A/B of the step.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call6; mkdir -p $O
echo "== kernel + short engine tests"
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_causal_gpu.py tests/test_streaming_gpu.py -q -m gpu -x -k "attention or bf16" 2>&1 | tail -4
B="--steps 4 --warmup 1 --no-diarization --traffic off --cpu-baseline-chunks 0"
pick() { grep -o "\"ms_per_step\": [0-9.]*\|\"gemm\": [0-9.]*\|\"attention\": [0-9.]*\|\"frac\": [0-9.]*" | tr "\n" " "; echo; }
for f in 0 1 0 1; do
  echo -n "RVB_ATTN_FOLD=$f: "
  RVB_ATTN_FOLD=$f timeout 150 python bench.py $B 2>/dev/null | tee $O/bench_fold$f.json | pick
done
echo "== long-form parity (bf16 on the bench hour, from PCM) with the folded form"
rm -f gpurun_out/parity_metrics.jsonl
timeout 600 python -m pytest tests/test_longform_gpu.py -q -m gpu -x -k "bf16" 2>&1 | tail -4
grep -h "r640_1h" gpurun_out/parity_metrics.jsonl | cut -c1-400
