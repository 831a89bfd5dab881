#!/bin/bash
# what the small-block comparison kernels really move: FETCH_SIZE per launch of the batch SAD / SATD kernels at random block positions
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2pp
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --prims --only compare --no-cpu > "$OUT/prims.txt" 2> "$OUT/err.txt"
true
cd "$GRAFT_REPO_ROOT"
python tools/rocprof_summary.py pmc $(find "$OUT/pmc_fetch" -name "*.db" | head -1) > "$OUT/pmc.txt" 2>&1
timeout 200 python bench.py --prims --only compare --no-cpu > "$OUT/prims_clean.txt" 2>/dev/null; cut -c1-170 "$OUT/prims_clean.txt" | head -26
cat "$OUT/pmc.txt" | cut -c1-200 | head -40
find "$OUT" -name '*.db' -delete
