#!/bin/bash
# Round 6: with the dependency-free frame-cost launch, is the lookahead seam worth it at 1080p (BASELINE configs[1], where every seam was gated off by size)?
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=${ENCODER_BENCH_NO_MD5:-}
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    la=s.get('lookahead_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'md5_equal', d.get('md5_equal_to_c_table'), 'estimates', la.get('frame_cost_estimates_served'), 'flat/walk/split', la.get('launches_flat_walk_split'), flush=True)
"; }
COMMON="--frame-threads 3 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-no-sad --seam-slots 24"
for r in 1 2; do
  run "cfg2 r$r control                 " --configs cfg2 --tables csplit --frames 96 --frame-threads 3 --seam-lookahead
  run "cfg2 r$r gated (round 5)         " --configs cfg2 --tables seam --frames 96 $COMMON --seam-lookahead
  run "cfg2 r$r lookahead seam, any size" --configs cfg2 --tables seam --frames 96 $COMMON --seam-lookahead --seam-lookahead-min-blocks 0
  export ENCODER_BENCH_NO_MD5=1
done
