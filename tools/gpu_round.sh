#!/bin/bash
# One GPU-box visit: parity suite, bench, rocprofv3 kernel stats and the two PMC traffic passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag]
# Outputs go under gpurun_out/<tag>/ ; convert the .db files with tools/rocprof_summary.py afterwards.
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; cat "$OUT/bench.json"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
cd "$ROOT"
python tools/rocprof_summary.py kernel-trace $(find "$OUT/stats" -name '*.db' | head -1) > "$OUT/kernel_stats.txt" 2>&1 || true
python tools/rocprof_summary.py pmc $(find "$OUT/pmc_fetch" -name '*.db' | head -1) $(find "$OUT/pmc_write" -name '*.db' | head -1) > "$OUT/pmc.txt" 2>&1 || true
find "$OUT" -name '*.db' -size +20M -delete
cat "$OUT/kernel_stats.txt" | head -30
grep x265hip "$OUT/pmc.txt"
ls -la "$OUT"
