#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2) rocpd SQLite outputs into the small text files kept under profiles/.

  python tools/rocprof_summary.py kernel-trace gpurun_out/prof/kt/me_results.db   > profiles/rNN_kernel_stats.txt
  python tools/rocprof_summary.py pmc gpurun_out/prof/pmc_fetch/me_results.db ...  > profiles/rNN_pmc.txt
"""
import sqlite3
import sys


def short(name, n=90):
    return name if len(name) <= n else name[:n - 3] + "..."


def kernel_trace(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':92s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
    for name, calls, tot, avg, pct in rows:
        print(f"{short(name):92s} {calls:6d} {tot / 1.0:12.2f} {avg:10.3f} {pct:7.2f}")


def pmc(paths):
    print("# rocprofv3 --pmc summary (one counter per pass; values per dispatch, KB for FETCH_SIZE / WRITE_SIZE)")
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                           "from counters_collection group by kernel_name, counter_name").fetchall()
        print(f"## {path}")
        print(f"{'kernel':92s} {'counter':>12s} {'n':>4s} {'avg':>16s} {'min':>16s} {'max':>16s} {'avg_dur_ns':>12s}")
        for k, c, n, a, mn, mx, d in rows:
            print(f"{short(k):92s} {c:>12s} {n:4d} {a:16.2f} {mn:16.2f} {mx:16.2f} {d:12.0f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernel-trace":
        kernel_trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])
