#!/bin/bash
# kernel trace of the banded step with the fused SAO launches (X265HIP_FUSE_SAO=1) and without
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for fuse in 1 0; do
  mkdir -p gpurun_out/r2bp$fuse
  cd /tmp
  X265HIP_FUSE_SAO=$fuse timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2bp$fuse/stats" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-encoder --banded --band-rows 4 --band-graphs 0 > "$GRAFT_REPO_ROOT/gpurun_out/r2bp$fuse/bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2bp$fuse/err.txt"
  cd "$GRAFT_REPO_ROOT"
  python tools/rocprof_summary.py kernel-trace $(find gpurun_out/r2bp$fuse/stats -name '*.db' | head -1) > gpurun_out/r2bp$fuse/kernel_stats.txt 2>&1
  echo "== fuse=$fuse"; head -14 gpurun_out/r2bp$fuse/kernel_stats.txt | cut -c1-150
  python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/r2bp$fuse/stats/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = c.execute(f"select start, end from {kd} order by start").fetchall()
# last 60% of the dispatches = the timed loop; busy time vs span
rows = rows[len(rows) * 4 // 10:]
span = rows[-1][1] - rows[0][0]
busy, cur_end = 0, rows[0][0]
for s, e in rows:
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print("dispatches", len(rows), "span_ms", span / 1e6, "gpu_busy_ms", busy / 1e6, "sum_kernel_ms", sum(e - s for s, e in rows) / 1e6)
PY
  find gpurun_out/r2bp$fuse -name '*.db' -delete
done
