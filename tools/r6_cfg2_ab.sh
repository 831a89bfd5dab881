#!/bin/bash
# Round 6 (round-5 verdict, next 7): is 1080p preset medium (BASELINE configs[1]) worth a GPU with the cost tables?  One box, interleaved: the host-only control, round 5's leg
# (every search seam gated off by size: 510 CTUs < 1000), and the cost tables ALONE with the size gate lifted (--subme 2: 21 positions, luma only, squares: 4.4 KB of records per CTU).
ROUNDS=${1:-2}
export ENCODER_BENCH_NO_MD5=${ENCODER_BENCH_NO_MD5:-}
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    c=s.get('cost_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'md5_equal', d.get('md5_equal_to_c_table'), 'cost_served', c.get('comparisons_served_from_records'),
          'share', c.get('served_share_of_satd_comparisons_with_context'), 'late', c.get('passed_on_records_not_arrived'), 'pairs', c.get('pairs_opened'), 'busy_ms', c.get('worker_busy_ms'), flush=True)
"; }
COMMON="--frame-threads 3 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-lookahead --seam-aq --seam-weight-analyse --seam-no-sad --seam-slots 24"
for r in $(seq 1 $ROUNDS); do
  run "cfg2 r$r control      " --configs cfg2 --tables csplit --frames 96 --frame-threads 3 --seam-lookahead
  run "cfg2 r$r r5 (gated)   " --configs cfg2 --tables seam --frames 96 $COMMON --seam-subpel
  run "cfg2 r$r cost1 only   " --configs cfg2 --tables seam --frames 96 $COMMON --seam-cost --seam-min-ctus 0
  run "cfg2 r$r cost1s4 only " --configs cfg2 --tables seam --frames 96 $COMMON --seam-cost --seam-cost-set-subme 4 --seam-min-ctus 0
done
