#!/bin/bash
# the two host-pointer stage services added late in round 4 - x265hip_aq_frame_host (calcAdaptiveQuantFrame) and x265hip_weight_analyse_host
# (weightAnalyse) - on and off in the real encoder, interleaved on one box: cfg3f (4K 8-bit slow star -F 5 on a FADE, --weightp on: weightAnalyse
# gets past its early exits on every P slice) and cfg3 (constant brightness: weightAnalyse exits early, only the AQ pass differs)
SE="--frame-threads 5 --seam-subpel-slots 12 --seam-streamed --seam-min-level 1 --seam-min-pu 16 --seam-lookahead --seam-subpel --seam-layout planes --seam-centre-range 57 --seam-range 12"
run() { python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | python -c "
import sys,json
for l in sys.stdin:
    tag=l.split(':')[0]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    print(tag, 'fps', d['fps'], 'cpu', d.get('process_cpu_seconds'), 'aq', s.get('aq_seam',{}).get('pictures_served'), 'wa', s.get('weight_analyse_seam',{}).get('slices_served'), 'with weight', s.get('weight_analyse_seam',{}).get('served_slices_with_a_weight'), 'md5', d.get('md5', '')[:8])"; }
for round in 1 2; do
  echo "cfg3f search seams only";            run --configs cfg3f --tables seam --frames 32 --seam-slots 24 $SE
  echo "cfg3f + aq + weightAnalyse";         run --configs cfg3f --tables seam --frames 32 --seam-slots 24 $SE --seam-aq --seam-weight-analyse
  echo "cfg3 search seams only";             run --configs cfg3 --tables seam --frames 48 --seam-slots 24 $SE
  echo "cfg3 + aq + weightAnalyse";          run --configs cfg3 --tables seam --frames 48 --seam-slots 24 $SE --seam-aq --seam-weight-analyse
done
echo "cfg3f C table (the baseline both are compared with)"; run --configs cfg3f --tables c --frames 32 --frame-threads 5
