#!/bin/bash
# Round 6 diagnostic: how much of the cfg3 leg is the encoder's start-up (the lookahead fills before the first picture is coded)?  The same leg at 24 / 48 / 96 frames, control and seams.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1 GPU_MAX_HW_QUEUES=16
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-aq --seam-weight-analyse --seam-lookahead"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
for n in ${FRAMES:-24 48 96}; do
  for t in csplit seam; do
    python tools/encoder_bench.py --configs cfg3 --tables $t --frames $n $COMMON $ARGS 2>&1 | grep "^\[encoder\]" | grep " $t:" | N=$n T=$t python -c "
import sys,json,os
for l in sys.stdin:
    d=json.loads(l.split(': ',1)[1])
    print('frames', os.environ['N'], os.environ['T'], 'fps', d['fps'], 'seconds', d['seconds'], 'cpu_s', d.get('process_cpu_seconds'), flush=True)"
  done
done
