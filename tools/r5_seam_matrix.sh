#!/bin/bash
# Round 5: how should the encoder legs configure the SAD seam?  (round-4 verdict, next 4: by what was measured, not by how many lookups get served.)
# ONE box, interleaved rounds, everything against the HOST-ONLY control (csplit): per configuration
#   control      C table with split sad_x3 / sad_x4, no GPU
#   all_l1       every seam, SAD planes of the 16x16-and-up levels        (round 4's legs)
#   all_l2       every seam, SAD planes of the 32x32 / 64x64 levels only  (40 % of the download)
#   no_sad       every seam but the SAD lookups
#   bash tools/r5_seam_matrix.sh [rounds] [configs]
ROUNDS=${1:-2}; CFGS=${2:-"cfg3 cfg3f cfg4"}
export ENCODER_BENCH_NO_MD5=1          # the bench legs hold the md5 check; here every leg would pay a C-table encode for it
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-subpel-slots 12 --seam-split-rest --seam-subpel --seam-lookahead --seam-aq --seam-weight-analyse"
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'hit', s.get('lookup_hit_rate'), 'served', s.get('lookups_served'), 'GB_down', round((s.get('bytes_downloaded',0)+s.get('subpel_seam',{}).get('bytes_downloaded',0))/1e9,2), 'md5_equal', d.get('md5_equal_to_c_table'))"; }
for r in $(seq 1 $ROUNDS); do
  for cfg in $CFGS; do
    case $cfg in cfg3) NF=48 SLOTS=24 ;; cfg3f) NF=24 SLOTS=24 ;; *) NF=24 SLOTS=40 ;; esac
    run "$cfg r$r control" --configs $cfg --tables csplit --frames $NF --frame-threads 5 --seam-lookahead
    run "$cfg r$r all_l1 " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON --seam-min-level 1
    run "$cfg r$r all_l2 " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON --seam-min-level 2
    run "$cfg r$r no_sad " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON --seam-min-level 1 --seam-no-sad
  done
done
