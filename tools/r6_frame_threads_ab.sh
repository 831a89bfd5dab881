cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export ENCODER_BENCH_NO_MD5=1
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    c=s.get('cost_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'cost_share', c.get('served_share_of_satd_comparisons_with_context'), 'late', c.get('passed_on_records_not_arrived'), flush=True)
"; }
COMMON="--seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-split-rest --seam-lookahead --seam-aq --seam-weight-analyse"
ARGS="--seam-slots 24 --seam-no-sad --seam-min-level 1 --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4"
for F in 5 6 8; do
  run "cfg3 F$F control" --configs cfg3 --tables csplit --frames 48 --frame-threads $F --seam-lookahead
  run "cfg3 F$F seams  " --configs cfg3 --tables seam --frames 48 --frame-threads $F $COMMON $ARGS
done
