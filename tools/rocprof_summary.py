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


def timeline(path, anchor="me_ctu"):
    """Dispatch timeline of ONE steady-state step: every kernel between the start of the last-but-one launch whose name contains `anchor`
    and the start of the last one - start / end relative to the anchor's start, queue, and the device idle time before each dispatch."""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute(f"select d.start, d.end, s.kernel_name, d.{q} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    anchors = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(anchors) < 3:
        print(f"fewer than three launches of {anchor}"); return
    # the middle of the run: bench.py's timed loop (the first launches are warm-up, the last ones its one-stream stage-timing pass)
    a0, a1 = anchors[len(anchors) // 2 - 1], anchors[len(anchors) // 2]
    t0 = rows[a0][0]
    print(f"# one step of {path}: from the start of one {anchor} launch to the start of the next = {(rows[a1][0] - t0) / 1e3:.1f} us")
    print(f"{'start_us':>9s} {'end_us':>9s} {'dur_us':>8s} {'idle_before_us':>14s} {'queue':>6s}  kernel")
    busy_until = t0
    for s_, e_, n, qq in rows[a0:a1 + 1]:
        idle = max(0, s_ - busy_until)
        name = n.split("(")[0].replace("void ", "").replace("x265hip::", "")
        print(f"{(s_ - t0) / 1e3:9.1f} {(e_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f} {idle / 1e3:14.1f} {str(qq):>6s}  {short(name, 80)}")
        busy_until = max(busy_until, e_)


if __name__ == "__main__":
    if sys.argv[1] == "kernel-trace":
        kernel_trace(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(*sys.argv[2:4])
    else:
        pmc(sys.argv[2:])
