#!/bin/bash
# sad_x3 / sad_x4 stubs of the test binding: one batched host call when no candidate lies in the window, (X265REF_SEAM_BATCH_MISS=1) against N single SADs (the default), on the
# fade (0.11 of the lookups hit: the unweighted references' searches wander over +-57) and at constant brightness (0.93 hit); interleaved on one box
SE="--frame-threads 5 --seam-subpel-slots 12 --seam-streamed --seam-min-level 1 --seam-min-pu 16 --seam-lookahead --seam-subpel --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-aq --seam-weight-analyse"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'served', s.get('lookups_served'), 'outside', s.get('outside_window'), 'md5', d.get('md5', '')[:8])"; }
for round in 1 2; do
  echo "cfg3f batched misses";  X265REF_SEAM_BATCH_MISS=1 run --configs cfg3f --tables seam --frames 24 --seam-slots 24 $SE
  echo "cfg3f single SADs";     run --configs cfg3f --tables seam --frames 24 --seam-slots 24 $SE
done
echo "cfg3 batched misses";  X265REF_SEAM_BATCH_MISS=1 run --configs cfg3 --tables seam --frames 48 --seam-slots 24 $SE
echo "cfg3 single SADs";     run --configs cfg3 --tables seam --frames 48 --seam-slots 24 $SE
