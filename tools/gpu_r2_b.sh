#!/bin/bash
# Round-2 GPU visit B: full parity suite (incl. the seam + me_cache tests), encoder-level legs with the stage-level seam.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -22 "$OUT/pytest.log"
EB="python tools/encoder_bench.py"
timeout 300 $EB --configs cfg2 --tables c,seam --frames 12 > "$OUT/enc_cfg2_seam.json" 2> "$OUT/enc_cfg2_seam.err"; echo "cfg2 seam rc=$?"; grep "^\[enc" "$OUT/enc_cfg2_seam.err"
timeout 300 $EB --configs cfg2 --tables c,seam --frames 6 --seam-verify > "$OUT/enc_cfg2_seam_verify.json" 2> "$OUT/enc_cfg2_seam_verify.err"; echo "cfg2 verify rc=$?"; grep "^\[enc" "$OUT/enc_cfg2_seam_verify.err"
timeout 400 $EB --configs cfg3 --tables c,seam --frames 6 > "$OUT/enc_cfg3_seam.json" 2> "$OUT/enc_cfg3_seam.err"; echo "cfg3 seam rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam.err"
timeout 400 $EB --configs cfg3 --tables seam --frames 6 --seam-min-pu 32 > "$OUT/enc_cfg3_seam32.json" 2> "$OUT/enc_cfg3_seam32.err"; echo "cfg3 seam32 rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam32.err"
timeout 400 $EB --configs cfg3 --tables seam --frames 6 --seam-range 57 --seam-min-pu 16 > "$OUT/enc_cfg3_seam57.json" 2> "$OUT/enc_cfg3_seam57.err"; echo "cfg3 seam57 rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_seam57.err"
timeout 500 $EB --configs cfg3 --tables c,hip --frames 2 --budget-s 0 > "$OUT/enc_cfg3_hip.json" 2> "$OUT/enc_cfg3_hip.err"; echo "cfg3 hip rc=$?"; grep "^\[enc" "$OUT/enc_cfg3_hip.err"
ls -la "$OUT"
