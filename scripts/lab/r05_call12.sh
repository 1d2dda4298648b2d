#!/bin/bash
# Round 5, GPU call 12: the new tests (joint_decoding tie retry, compute_feats with other settings) + the diarization kernel tests once more.
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_joint_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu -k "joint or compute_feats or fbank or decode" 2>&1 | tail -15
