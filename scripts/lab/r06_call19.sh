#!/bin/bash
# round 6, call 19: per-dispatch durations of the segmentation network's launches inside one diarization step (why do the LSTM
# projections take 1.6 ms in the engine and 0.65-0.97 ms in the kernel benchmark?)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call19; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench_diar.py --steps 1 --warmup 1 --traffic off --cpu-baseline-windows 0 > $O/stdout.log 2>&1 < /dev/null
T=$(ls $O/tr/*/*kernel_trace.csv | head -1)
python - "$T" <<'PY' | tee $O/segmentation_dispatches.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rvb::", "")[:60]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last window_stats dispatch opens the last step's segmentation
idx = [i for i, r in enumerate(rows) if "window_stats" in r[2]]
i0 = idx[-1]
prev_end = rows[i0][0]
for s, e, n in rows[i0:i0 + 24]:
    print(f"{n:62s} start +{(s - rows[i0][0]) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  duration {(e - s) / 1e3:9.1f} us")
    prev_end = e
PY
rm -rf $O/tr
