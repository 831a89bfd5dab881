#!/bin/bash
# Round 6 diagnostic: does the real encode's fps follow the LATENCY of a frame-cost estimate?  X265REF_LA_DELAY_US (test hook of the binding) adds a sleep to every served estimate.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-aq --seam-weight-analyse --seam-lookahead"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
for d in ${DELAYS:-0 2000 5000 0}; do
  X265REF_LA_DELAY_US=$d python tools/encoder_bench.py --configs cfg3 --tables seam --frames 48 $COMMON $ARGS 2>&1 | grep "^\[encoder\]" | grep " seam:" | D=$d python -c "
import sys,json,os
for l in sys.stdin:
    d=json.loads(l.split(': ',1)[1]); la=d.get('seam',{}).get('lookahead_seam',{})
    print('delay_us', os.environ['D'], 'fps', d['fps'], 'seconds', d['seconds'], 'cpu_s', d.get('process_cpu_seconds'), 'estimates', la.get('frame_cost_estimates_served'), 'flat/walk/split', la.get('launches_flat_walk_split'), flush=True)"
done
