#!/bin/bash
# Round 6 diagnostic: the frame-cost estimates by kind (X265HIP_LA_STATS=1: what each had to search, wall time per call) under 4 / 16 / 32 hardware queues
# (GPU_MAX_HW_QUEUES: the lookahead's pool threads call in concurrently, each on its own stream).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1 X265HIP_LA_STATS=1
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-aq --seam-weight-analyse --seam-lookahead"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
for q in ${QUEUES:-4 16 32 4 16}; do
  echo "== GPU_MAX_HW_QUEUES=$q ${EXTRA_ENV:-}"
  env GPU_MAX_HW_QUEUES=$q ${EXTRA_ENV:-} python tools/encoder_bench.py --configs cfg3 --tables seam --frames 48 $COMMON $ARGS 2>&1 | grep -E "^\[encoder\].* seam:|lowres_cost_host" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('libx265hip'): print(l.rstrip()); continue
    d=json.loads(l.split(': ',1)[1]); la=d.get('seam',{}).get('lookahead_seam',{})
    print('fps', d['fps'], 'seconds', d['seconds'], 'cpu_s', d.get('process_cpu_seconds'), 'estimates', la.get('frame_cost_estimates_served'), 'flat/walk/split', la.get('launches_flat_walk_split'), flush=True)"
done
