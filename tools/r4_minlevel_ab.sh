#!/bin/bash
# do the 16x16 rasters earn their bytes?  planes layout with min_level 1 (16x16 / 32x32 / 64x64 rasters, min_pu 16) against min_level 2
# (32x32 / 64x64 rasters only, min_pu 32: 52 -> 20 bytes per candidate), interleaved on one box; bytes downloaded from the stream's own report
SE="--frame-threads 5 --seam-subpel-slots 12 --seam-streamed --seam-lookahead --seam-subpel --seam-layout planes --seam-centre-range 57 --seam-range 12"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'served', s.get('lookups_served'), 'ctx', s.get('calls_with_lookup_context'), 'MB down', round(s.get('bytes_downloaded', 0)/1e6), 'md5', d.get('md5', '')[:8])"; }
for round in 1 2; do
  for m in 1 2; do echo "cfg3 min_level $m"; run --configs cfg3 --tables seam --frames 48 --seam-slots 24 --seam-min-level $m --seam-min-pu $((16 * m)) $SE; done
  for m in 1 2; do echo "cfg4 min_level $m"; run --configs cfg4 --tables seam --frames 24 --seam-slots 40 --seam-min-level $m --seam-min-pu $((16 * m)) $SE; done
done
