#!/bin/bash
# Where a step of lowres_cost_kernel goes (measurement aid): duration + instruction counters of ONE frame cost estimate at a given source size.
# bash tools/lowres_cost_counters.sh <tag> [width height]
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/${1:-lrc}; W=${2:-1920}; H=${3:-1080}
mkdir -p "$OUT"; export TMPDIR=/tmp
cat > "$OUT/one.py" <<PY
import importlib, sys, os, torch
sys.path.insert(0, "$PWD")
F = importlib.import_module("x265-yuuki-asuna_amd.frames"); P = importlib.import_module("x265-yuuki-asuna_amd.pipeline"); S = importlib.import_module("x265-yuuki-asuna_amd.stages")
dev = torch.device("cuda:0")
clip = F.synth_clip($W, $H, 2, depth=8, seed=5)
cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
lc, lr = S.Lookahead($W, $H, 8, dev), S.Lookahead($W, $H, 8, dev)
lc.run(cur); lr.run(ref)
st = [S.LookaheadCost(lc, dev)]
for _ in range(4):
    S.LookaheadCost.run_batch(st, [lc], [lr])
torch.cuda.synchronize()
print("blocks", lc.wcu, lc.hcu, "steps", lc.wcu + 2 * (lc.hcu - 1))
PY
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/t" -o b -- python "$OLDPWD/$OUT/one.py" > "$OLDPWD/$OUT/t.out" 2> "$OLDPWD/$OUT/t.err" )
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
    i=$((i + 1))
    ( cd /tmp && rocprofv3 --pmc $grp --kernel-trace -d "$OLDPWD/$OUT/p$i" -o b -- python "$OLDPWD/$OUT/one.py" > /dev/null 2> "$OLDPWD/$OUT/p$i.err" )
done
cat "$OUT/t.out"
python - "$OUT" <<'PY'
import glob, sqlite3, sys
out = sys.argv[1]
for db in glob.glob(out + "/t/**/*.db", recursive=True):
    c = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if kd and ks:
        for r in c.execute(f"select s.kernel_name, count(*), avg(d.end - d.start) from {kd[0]} d join {ks[0]} s on d.kernel_id = s.id group by s.kernel_name"):
            if "lowres_cost" in r[0]: print("duration", r[0][:60], "calls", r[1], "avg us", r[2] / 1e3)
for db in glob.glob(out + "/p*/**/*.db", recursive=True):
    c = sqlite3.connect(db).cursor()
    try:
        for k, n, cnt, v in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            if "lowres_cost" in k: print("counter", n, f"{v:.0f}", "dispatches", cnt)
    except Exception as e:
        print("unreadable", db, e)
PY
find "$OUT" -name '*.db' -delete
