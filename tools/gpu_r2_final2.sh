#!/bin/bash
# Round-2 closing GPU visit (second): full parity suite, smoke, default bench line, rocprofv3 kernel stats of the same command
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2w
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest.log" | tail -30
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-600 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
cd "$ROOT"
python tools/rocprof_summary.py kernel-trace $(find "$OUT/stats" -name '*.db' | head -1) > "$OUT/kernel_stats.txt" 2>&1 || true
find "$OUT" -name '*.db' -size +20M -delete
head -30 "$OUT/kernel_stats.txt"
