#!/bin/bash
# Round 6 diagnostic: the seam leg three times in ONE process (cfg3, 48 frames): what the first encode pays once per process (HIP start-up, code objects, first allocations)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1 GPU_MAX_HW_QUEUES=16 X265HIP_LA_STATS=1
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-aq --seam-weight-analyse --seam-lookahead"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
python tools/encoder_bench.py --configs cfg3 --tables ${TABLES:-seam,seam,seam} --frames ${FRAMES:-48} $COMMON $ARGS 2>&1 | grep -E "^\[encoder\].* seam:|lowres_cost_host" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('libx265hip'): print(l.rstrip()); continue
    d=json.loads(l.split(': ',1)[1])
    print('fps', d['fps'], 'seconds', d['seconds'], 'wall with open / close', d.get('wall_seconds_with_open_close'), 'cpu_s', d.get('process_cpu_seconds'), flush=True)"
