#!/bin/bash
# which partitions are worth a lookup?  --seam-min-pu 16 against 32 (16x16-level partitions answered by the host's own SAD), interleaved on one box
SE="--frame-threads 5 --seam-subpel-slots 12 --seam-streamed --seam-min-level 1 --seam-lookahead --seam-subpel --seam-layout planes --seam-centre-range 57 --seam-range 12"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'served', s.get('lookups_served'), 'ctx', s.get('calls_with_lookup_context'))"; }
for round in 1 2; do
  for m in 16 32; do echo "cfg3 min_pu $m"; run --configs cfg3 --tables seam --frames 48 --seam-slots 24 --seam-min-pu $m $SE; done
  for m in 16 32; do echo "cfg4 min_pu $m"; run --configs cfg4 --tables seam --frames 24 --seam-slots 40 --seam-min-pu $m $SE; done
done
