#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports: calls, total / average duration, share of GPU time.
usage: python scripts/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    unit = 1e-3   # rocpd durations are ns in `kernels`; top_kernels reports microseconds
    lines = ["| kernel | calls | total (us) | avg (us) | % of kernel time |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"| `{name}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
    gem = [(c, t) for n, c, t, a, p in rows if "gemm_kernel" in n]
    if gem:
        c = sum(x[0] for x in gem); t = sum(x[1] for x in gem)
        lines.append("")
        lines.append(f"all `rvb::gemm_kernel` instantiations: {c} launches, {t:.1f} us total, **{t / c:.2f} us average**")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
