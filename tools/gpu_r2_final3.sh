#!/bin/bash
# refresh of the round-2 closing artefacts after the last host-side changes: default bench line + kernel stats of the same command, banded line, soak
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2x
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json"; tail -2 "$OUT/bench.err" | cut -c1-300
timeout 300 python bench.py --banded --no-encoder --no-cpu-baseline --steps 60 > "$OUT/bench_banded.json" 2>/dev/null; echo "banded rc=$?"; cut -c1-300 "$OUT/bench_banded.json"
timeout 200 python tools/band_soak.py 40 23 2>&1 | tail -1
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
cd "$ROOT"
python tools/rocprof_summary.py kernel-trace $(find "$OUT/stats" -name '*.db' | head -1) > "$OUT/kernel_stats.txt" 2>&1 || true
find "$OUT" -name '*.db' -delete
head -12 "$OUT/kernel_stats.txt" | cut -c1-150
