#!/bin/bash
# Round 6 diagnostic: is the real encode at cfg3 bound by its LOOKAHEAD (slice-type decision: b-adapt 2 over 25 pictures, ~20 frame-cost estimates per picture, each a
# latency-bound launch through the lookahead seam) rather than by the frame encoders?  The same legs with a cheaper lookahead: if the fps jumps, it is.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    la=s.get('lookahead_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'frame_cost_estimates_served', la.get('frame_cost_estimates_served'), flush=True)
"; }
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-aq --seam-weight-analyse"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
for extra in "" "b-adapt=0" "b-adapt=0,rc-lookahead=5" ; do
  export ENCODER_BENCH_EXTRA_OPTS="$extra"
  run "cfg3 [$extra] control                " --configs cfg3 --tables csplit --frames 48 --frame-threads 5 --seam-lookahead
  run "cfg3 [$extra] seams + lookahead seam " --configs cfg3 --tables seam --frames 48 $COMMON $ARGS --seam-lookahead
  run "cfg3 [$extra] seams, HOST lookahead  " --configs cfg3 --tables seam --frames 48 $COMMON $ARGS
done
