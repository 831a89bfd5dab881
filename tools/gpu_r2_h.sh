#!/bin/bash
# Round-2 GPU visit H: parity suite incl. the banded pipeline; what the band granularity costs on one GPU (bench.py --banded).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2h}
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed|Error" "$OUT/pytest.log" | tail -30
for br in 4 8 2; do
timeout 600 python bench.py --banded --band-rows $br --no-cpu-baseline > "$OUT/bench_banded_$br.json" 2> "$OUT/bench_banded_$br.err"; echo "banded $br rc=$?"
python - "$OUT/bench_banded_$br.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["parallelism"][:60], d["config"]["checksum"])
PY
tail -2 "$OUT/bench_banded_$br.err"
done
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"; python - "$OUT/bench_plain.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["checksum"])
PY
