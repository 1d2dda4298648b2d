"""Where the device idles inside a bench step: reads a `rocprofv3 --kernel-trace --output-format csv` file, orders the
dispatches by start time and reports the gaps between the end of one kernel and the start of the next (over all queues: a gap
counts only when NO kernel is running), summed by (kernel before, kernel after).

    rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --steps 2 --warmup 1 ...
    python scripts/gap_report.py out/*/*_kernel_trace.csv [min_gap_us [marker_kernel [skip]]]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
    rows.sort()
    # optional: start at the (skip + 1)-th dispatch of a marker kernel (model loading and warm-up steps come before)
    if len(sys.argv) > 3:
        marker, skip = sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0
        hits = [i for i, r in enumerate(rows) if marker in r[2]]
        rows = rows[hits[skip]:]
        print(f"window: from dispatch {skip + 1} of {marker!r} ({len(hits)} in the trace)")
    busy_end, prev = rows[0][1], rows[0][2]
    gaps = collections.defaultdict(lambda: [0.0, 0])
    total_gap, total_busy, t_first = 0.0, 0.0, rows[0][0]
    big = []
    for s, e, name in rows[1:]:
        if s > busy_end:
            g = (s - busy_end) / 1e3
            total_gap += g
            if g >= min_gap:
                k = (prev, name)
                gaps[k][0] += g
                gaps[k][1] += 1
                big.append((g, prev, name))
        if e > busy_end:
            busy_end, prev = e, name
    span = (busy_end - t_first) / 1e3
    print(f"{len(rows)} dispatches over {span / 1e3:.1f} ms; device idle {total_gap / 1e3:.2f} ms in total "
          f"({100 * total_gap / span:.1f} %), gaps >= {min_gap:g} us:")
    for (a, b), (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {g / 1e3:8.2f} ms  x{n:<5d} after {a}\n{'':22s}before {b}")
    print("largest single gaps:")
    for g, a, b in sorted(big, reverse=True)[:12]:
        print(f"  {g / 1e3:8.3f} ms  {a}  ->  {b}")


if __name__ == "__main__":
    main()
