#!/bin/bash
# sample the GPU clock / power while a GEMM loop runs (is the plateau a power/clock limit?)
export PYTHONPATH=$GRAFT_REPO_ROOT
( for i in $(seq 1 12); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clocks.log 2>&1 &
SMI=$!
sleep 1
DTYPES=bf16 timeout 60 python - <<'PY'
import ctypes as C, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from reverb_amd import _lib
lib = _lib.load()
ms, md = C.c_double(0), C.c_double(0)
t0 = time.time()
while time.time() - t0 < 4.0:
    lib.rvb_test_gemm_bench(1, 90112, 1024, 19456, 2, 20, 0, 1, 0, C.byref(ms), C.byref(md))
print("gemm", ms.value, "ms", 2.0 * 90112 * 1024 * 19456 / ms.value / 1e9, "TF/s")
PY
wait $SMI
cat gpurun_out/clocks.log
