#!/bin/bash
# Round-2 GPU visit D: parity suite after the ME emission trims / wave-per-CTU sao_decide, the default bench line incl. the encoder-level leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2d}
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest.log" | tail -30
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ("value","ms_per_step","bit_exact")}, d["stages_ms"], d["roofline"]["frac"], d.get("encoder_summary"), d.get("encoder",{}).get("error"))
PY
tail -3 "$OUT/bench.err"
