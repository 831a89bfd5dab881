#!/bin/bash
# FIRST THING for the next round (left unmeasured when round 4's GPU budget ran out): every seam alone and together against the HOST-ONLY control (csplit), not against
# the C table - one box, interleaved.  cfg3 (8-bit) and cfg4 (10-bit); ~6 min of GPU time.
#   bash tools/r5_seam_matrix.sh [frames3] [frames4]
F3=${1:-48}; F4=${2:-24}
BASE="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-level 1 --seam-min-pu 16 --seam-subpel-slots 12 --seam-split-rest"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'served', s.get('lookups_served'), 'subpel', s.get('subpel_seam',{}).get('subpel_compares_served'), 'la', s.get('lookahead_seam',{}).get('frame_cost_estimates_served'), 'MB', round(s.get('bytes_downloaded',0)/1e6), 'md5', d.get('md5','')[:8])"; }
for cfg in cfg3 cfg4; do
  [ $cfg = cfg3 ] && NF=$F3 SLOTS=24 || NF=$F4 SLOTS=40
  echo "== $cfg: C table and host-only control (the legs below add --lookahead-slices 1 only when the lookahead seam is on: compare md5s within a group)"
  run --configs $cfg --tables c,csplit --frames $NF --frame-threads 5
  echo "== $cfg: SAD seam alone (16x16 and up)";              run --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $BASE
  echo "== $cfg: SAD seam alone, 32x32 and up (min_level 2)";  run --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS ${BASE/--seam-min-level 1 --seam-min-pu 16/--seam-min-level 2 --seam-min-pu 32}
  echo "== $cfg: sub-sample seam alone";                       run --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $BASE --seam-no-sad --seam-subpel
  echo "== $cfg: lookahead + AQ + weightAnalyse alone";        run --configs $cfg --tables c,csplit,seam --frames $NF --seam-slots $SLOTS $BASE --seam-no-sad --seam-lookahead --seam-aq --seam-weight-analyse
  echo "== $cfg: everything but the SAD seam";                 run --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $BASE --seam-no-sad --seam-subpel --seam-lookahead --seam-aq --seam-weight-analyse
  echo "== $cfg: everything";                                  run --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $BASE --seam-subpel --seam-lookahead --seam-aq --seam-weight-analyse
done
