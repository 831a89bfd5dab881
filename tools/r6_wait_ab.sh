#!/bin/bash
# Round 6: what do BLOCKING host waits (hipDeviceScheduleBlockingSync, the library's default from round 6 on) buy the real encode against the runtime's spinning waits
# (X265HIP_WAIT=spin)?  One box, interleaved, the bench's own leg configurations: cfg3 = cost tables alone (85-position set), cfg4 / cfg5 = SAD rasters 32x32-up + phase planes + tables.
ROUNDS=${1:-2}
CFGS=${2:-"cfg3 cfg4"}
export ENCODER_BENCH_NO_MD5=1
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    c=s.get('cost_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'cost_share', c.get('served_share_of_satd_comparisons_with_context'), 'late', c.get('passed_on_records_not_arrived'), 'busy_ms', c.get('worker_busy_ms'), flush=True)
"; }
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-lookahead --seam-aq --seam-weight-analyse"
for r in $(seq 1 $ROUNDS); do
  for cfg in $CFGS; do
    case $cfg in cfg3) NF=48 ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4" ;;
                 cfg5) NF=3  ARGS="--seam-slots 12 --seam-min-level 1 --seam-subpel --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4" ;;
                 *)    NF=24 ARGS="--seam-slots 40 --seam-min-level 2 --seam-subpel --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4" ;; esac
    run "$cfg r$r control      " --configs $cfg --tables csplit --frames $NF --frame-threads 5 --seam-lookahead
    X265HIP_WAIT=spin run "$cfg r$r seams, spin  " --configs $cfg --tables seam --frames $NF $COMMON $ARGS
    X265HIP_WAIT=block run "$cfg r$r seams, block " --configs $cfg --tables seam --frames $NF $COMMON $ARGS
  done
done
