#!/bin/bash
# Round-2 GPU visit G: full suite, default bench line (encoder leg with both seams), repeated cfg3 encoder legs for run-to-run statistics.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2g}
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed|Error" "$OUT/pytest.log" | tail -30
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ("value","ms_per_step","bit_exact")}, d["stages_ms"], d["roofline"]["frac"], d.get("encoder_summary"), d.get("encoder",{}).get("error"))
PY
grep "^\[enc" "$OUT/bench.err" | cut -c1-400
EB="python tools/encoder_bench.py"
for i in 1 2 3; do
timeout 400 $EB --configs cfg3 --tables c,seam --frames 12 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg3_la_$i.json" 2> "$OUT/enc_cfg3_la_$i.err"; echo "cfg3 la rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_la_$i.err" | cut -c1-200
done
timeout 400 $EB --configs cfg4 --tables c,seam --frames 6 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg4_la.json" 2> "$OUT/enc_cfg4_la.err"; echo "cfg4 la rc=$?"; grep "^\[enc" "$OUT/enc_cfg4_la.err" | cut -c1-300
