#!/bin/bash
# Round-2 GPU visit E: the lookahead seam (x265hip_lowres_cost_host behind CostEstimateGroup::estimateFrameCost) - parity tests + encoder legs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2e}
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|passed|failed|Error" "$OUT/pytest.log" | tail -30
EB="python tools/encoder_bench.py"
for i in 1 2; do
timeout 300 $EB --configs cfg2 --tables c,seam --frames 16 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg2_la_$i.json" 2> "$OUT/enc_cfg2_la_$i.err"; echo "cfg2 la rc=$?"; grep "^\[enc" "$OUT/enc_cfg2_la_$i.err"
timeout 400 $EB --configs cfg3 --tables c,seam --frames 10 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg3_la_$i.json" 2> "$OUT/enc_cfg3_la_$i.err"; echo "cfg3 la rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_la_$i.err"
done
timeout 400 $EB --configs cfg3 --tables c,seam --frames 10 --seam-range 24 --seam-min-pu 64 --seam-lookahead > "$OUT/enc_cfg3_laonly.json" 2> "$OUT/enc_cfg3_laonly.err"; echo "cfg3 la-only rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_laonly.err"
