#!/bin/bash
# Round-2 GPU visit C: full parity suite, default bench line (chroma + SAO in the loop, sign hiding), seam legs with batched / row-interleaved surfaces,
# rocprofv3 kernel stats + the two PMC traffic passes of the bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest.log" | tail -30
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
EB="python tools/encoder_bench.py"
timeout 300 $EB --configs cfg2 --tables c,seam --frames 12 > "$OUT/enc_cfg2_seam.json" 2> "$OUT/enc_cfg2_seam.err"; echo "cfg2 seam rc=$?"; grep "^\[enc" "$OUT/enc_cfg2_seam.err"
timeout 400 $EB --configs cfg3 --tables c,seam --frames 8 > "$OUT/enc_cfg3_seam.json" 2> "$OUT/enc_cfg3_seam.err"; echo "cfg3 seam rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam.err"
timeout 400 $EB --configs cfg3 --tables seam --frames 8 --seam-min-pu 16 > "$OUT/enc_cfg3_seam16.json" 2> "$OUT/enc_cfg3_seam16.err"; echo "cfg3 seam16 rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam16.err"
timeout 400 $EB --configs cfg3 --tables seam --frames 8 --seam-range 24 > "$OUT/enc_cfg3_seam_r24.json" 2> "$OUT/enc_cfg3_seam_r24.err"; echo "cfg3 seam r24 rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam_r24.err"
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
head -40 "$OUT/kernel_stats.txt"
grep x265hip "$OUT/pmc.txt" | head -20
