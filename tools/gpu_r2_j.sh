#!/bin/bash
# Round-2 GPU visit J: does a larger HW queue budget help the concurrent lookahead estimates?  (GPU_MAX_HW_QUEUES)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r2j}
mkdir -p "$OUT"
cd "$ROOT"
for q in 16 8; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/la_host_probe.py 2>&1 | grep "16 host\| 4 host" | tee "$OUT/probe_q$q.txt"
done
EB="python tools/encoder_bench.py"
GPU_MAX_HW_QUEUES=16 timeout 300 $EB --configs cfg2 --tables c,seam --frames 16 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg2_q16.json" 2> "$OUT/enc_cfg2_q16.err"; grep "^\[enc" "$OUT/enc_cfg2_q16.err" | cut -c1-120
GPU_MAX_HW_QUEUES=16 timeout 400 $EB --configs cfg3 --tables c,seam --frames 12 --seam-range 24 --seam-lookahead > "$OUT/enc_cfg3_q16.json" 2> "$OUT/enc_cfg3_q16.err"; grep "^\[enc" "$OUT/enc_cfg3_q16.err" | cut -c1-120
