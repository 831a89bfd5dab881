#!/bin/bash
# dispatch timeline of one band of the banded step, fused / per-plane SAO
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for fuse in 1 0; do
  mkdir -p gpurun_out/r2tl$fuse
  cd /tmp
  X265HIP_FUSE_SAO=$fuse timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/r2tl$fuse/stats" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-encoder --banded --band-rows 4 --band-streams 1 > "$GRAFT_REPO_ROOT/gpurun_out/r2tl$fuse/bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2tl$fuse/err.txt"
  cd "$GRAFT_REPO_ROOT"
  echo "== fuse=$fuse $(python -c "import json;d=json.loads(open('gpurun_out/r2tl$fuse/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
  python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/r2tl$fuse/stats/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
rows = c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# frame starts (lowres_init) and the largest idle gaps of the run
starts = [r[0] for r in rows if "lowres_init" in r[2]]
print("frame periods ms:", [round((b - a) / 1e6, 2) for a, b in zip(starts[:-1], starts[1:])])
gaps = []
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    if s - prev_end > 20000:
        gaps.append(((s - prev_end) / 1e3, prev_name.split("(")[0][-40:], n.split("(")[0][-40:], (s - rows[0][0]) / 1e6))
    if e > prev_end:
        prev_end, prev_name = e, n
for g in sorted(gaps, reverse=True)[:25]:
    print(f"gap {g[0]:9.1f} us at {g[3]:9.2f} ms after {g[1]} before {g[2]}")
PY
  find gpurun_out/r2tl$fuse -name '*.db' -delete
done
