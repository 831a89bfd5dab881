#!/bin/bash
# Instruction-mix counters of every kernel of the default bench step (measurement aid): one rocprofv3 --pmc pass per counter group, --kernel-trace only
# (never combined with the hip / hsa / memory-copy traces).  Prints per-kernel averages per dispatch and per wavefront.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/${1:-inst}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-encoder --no-verify"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA"; do
    i=$((i + 1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace -d "$OLDPWD/$OUT/p$i" -o b -- python "$OLDPWD/bench.py" $ARGS > /dev/null 2> "$OLDPWD/$OUT/p$i.err" )
done
python - "$OUT" <<'PY'
import glob, sqlite3, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(dict)
for db in glob.glob(out + "/p*/**/*.db", recursive=True):
    c = sqlite3.connect(db).cursor()
    try:
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print("unreadable", db, e); continue
    for k, n, cnt, v in rows:
        if "x265hip" in k:
            acc[k.split("(")[0].replace("void ", "").replace("x265hip::", "")][n] = v
names = sorted({n for v in acc.values() for n in v})
print("kernel".ljust(62), " ".join(n.replace("SQ_", "")[:12].rjust(13) for n in names))
for k, v in sorted(acc.items()):
    print(k[:62].ljust(62), " ".join((f"{v[n]:13.0f}" if n in v else " " * 13) for n in names))
print("# per wavefront (counter / SQ_WAVES)")
for k, v in sorted(acc.items()):
    w = v.get("SQ_WAVES")
    if w:
        print(k[:62].ljust(62), " ".join((f"{v[n] / w:13.1f}" if n in v else " " * 13) for n in names))
PY
find "$OUT" -name '*.db' -delete
