BASE="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-level 1 --seam-min-pu 16 --seam-subpel-slots 12 --seam-split-rest --seam-subpel --seam-lookahead --seam-aq --seam-weight-analyse --seam-slots 24"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'served', s.get('lookups_served'), 'md5', d.get('md5','')[:8])"; }
echo "everything but the SAD seam"; run --configs cfg3 --tables seam --frames 32 $BASE --seam-no-sad
echo "everything";                  run --configs cfg3 --tables seam --frames 32 $BASE
