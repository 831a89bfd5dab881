#!/bin/bash
# SQ counter pass over the bench (dominant kernel: me_ctu_q_kernel): how busy the VALU is, how long waves wait.
# Usage on the GPU box: bash tools/gpu_me_counters.sh [tag] -> gpurun_out/<tag>/me_counters.txt
set -u
TAG=${1:-mecnt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    -d "$OUT/pmc_sq" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
python - "$OUT" <<'PY' > "$OUT/me_counters.txt"
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for key in ("me_ctu_q_kernel", "subpel_refine_kernel", "sao_stats_kernel", "inter_recon_kernel"):
        if key in k:
            agg.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc SQ_* (one pass) over: bench.py --steps 6 --warmup 2 --no-cpu-baseline; per-dispatch averages")
for k, cs in agg.items():
    a = {n: sum(v) / len(v) for n, v in cs.items()}
    line = f"{k}: " + ", ".join(f"{n}={a[n]:.4g}" for n in sorted(a))
    if "SQ_ACTIVE_INST_VALU" in a and "SQ_BUSY_CYCLES" in a:
        # SQ_ACTIVE_INST_VALU counts cycles (x4 quad-cycles) the VALU executes, summed over SIMDs; SQ_BUSY_CYCLES per SE
        line += f" | VALU instr per wave = {a['SQ_INSTS_VALU'] / max(a['SQ_WAVES'], 1):.0f}, wait/wave-cycles = {a['SQ_WAIT_INST_ANY'] / max(a['SQ_WAVE_CYCLES'], 1):.3f}, active-valu/wave-cycles = {a['SQ_ACTIVE_INST_VALU'] / max(a['SQ_WAVE_CYCLES'], 1):.3f}"
    print(line)
PY
find "$OUT" -name '*.csv' -size +5M -delete
cat "$OUT/me_counters.txt"
