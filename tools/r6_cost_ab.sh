#!/bin/bash
# Round 6: what do the sub-sample COST TABLES (x265hip_cost_stream behind MotionEstimate::subpelCompare) add to the encoder legs?  ONE box, interleaved rounds,
# everything against the HOST-ONLY control (csplit):
#   control        C table with split sad_x3 / sad_x4, no GPU
#   r5             round 5's leg: phase planes + lookahead + AQ + weightAnalyse services (no SAD lookups at 8 bits, 32x32-and-up rasters above)
#   r5+cost1       the same + cost tables, 1 candidate vector per PU          (tables first, phase planes for what they do not answer)
#   r5+cost2       the same + cost tables, 2 candidate vectors per PU
#   cost1 only     cost tables WITHOUT the phase planes (nothing but records travels for the sub-sample half)
#   ...s4          the records hold the position set of --subme 4 (85 positions) although the encode runs --subme 3 (49): fractional predictors stay inside it
#   bash tools/r6_cost_ab.sh [rounds] [configs] [verify: 1 = every served value re-evaluated by the reference in flight, first round only]
ROUNDS=${1:-2}; CFGS=${2:-"cfg3"}; VERIFY=${3:-0}
COMMON="--frame-threads 5 --seam-streamed --seam-layout planes --seam-centre-range 57 --seam-range 12 --seam-min-pu 16 --seam-subpel-slots 12 --seam-split-rest --seam-lookahead --seam-aq --seam-weight-analyse"
run() { tag=$1; shift; python tools/encoder_bench.py "$@" 2>&1 | grep "^\[encoder\]" | TAG=$tag python -c "
import sys,json,os
for l in sys.stdin:
    leg=l.split(':')[0].split()[-1]; d=json.loads(l.split(': ',1)[1]); s=d.get('seam',{})
    if leg == 'c': continue
    c=s.get('cost_seam',{}); sp=s.get('subpel_seam',{})
    print(os.environ['TAG'], leg, 'fps', d['fps'], 'cpu_s', d.get('process_cpu_seconds'), 'md5_equal', d.get('md5_equal_to_c_table'),
          'GB_down', round((s.get('bytes_downloaded',0)+sp.get('bytes_downloaded',0)+c.get('bytes_downloaded',0))/1e9,2),
          'cost_served', c.get('comparisons_served_from_records'), 'share', c.get('served_share_of_satd_comparisons_with_context'), 'other', c.get('passed_on_other_vector_or_position'),
          'late', c.get('passed_on_records_not_arrived'), 'no_ctx', c.get('calls_without_context'), 'of', c.get('motion_estimate_calls_seen'), 'pairs', c.get('pairs_opened'), 'stale', c.get('stale_pairs'),
          'sad_served', c.get('sad_typed_comparisons_served_from_records'), 'failed', c.get('failed'), 'busy_ms', c.get('worker_busy_ms'), 'phase_served', sp.get('subpel_compares_served'), 'mismatch', c.get('verify_mismatches'), flush=True)
"; }
for r in $(seq 1 $ROUNDS); do
  for cfg in $CFGS; do
    case $cfg in cfg3) NF=48 SLOTS=24 SAD="--seam-no-sad --seam-min-level 1" ;; cfg3f) NF=24 SLOTS=24 SAD="--seam-no-sad --seam-min-level 1" ;; cfg5) NF=3 SLOTS=40 SAD="--seam-min-level 2" ;; *) NF=24 SLOTS=40 SAD="--seam-min-level 2" ;; esac
    V=""; [ "$VERIFY" = "1" ] && [ "$r" = "1" ] && V="--seam-verify"
    [ "$r" = "1" ] || export ENCODER_BENCH_NO_MD5=1          # the md5 comparison (a C-table encode per leg) in the first round only
    run "$cfg r$r control   " --configs $cfg --tables csplit --frames $NF --frame-threads 5 --seam-lookahead
    run "$cfg r$r r5        " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-subpel
    run "$cfg r$r r5+cost1  " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-subpel --seam-cost --seam-cost-candidates 1 $V
    run "$cfg r$r r5+cost2  " --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-subpel --seam-cost --seam-cost-candidates 2 $V
    run "$cfg r$r cost1 only" --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-cost --seam-cost-candidates 1
    if [ "$cfg" = "cfg3" ] || [ "$cfg" = "cfg3f" ]; then          # --subme 3 encodes: records with the 85 positions of workload row 4 (a superset of row 3's 49)
    run "$cfg r$r r5+cost1s4" --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-subpel --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4
    run "$cfg r$r cost1s4 only" --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4
    fi
    # ... and the SAD-typed comparisons (the predictor candidates of every search) from the same records
    run "$cfg r$r cost1s4+sad only" --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4 --seam-cost-sad $V
    run "$cfg r$r r5+cost1s4+sad" --configs $cfg --tables seam --frames $NF --seam-slots $SLOTS $COMMON $SAD --seam-subpel --seam-cost --seam-cost-candidates 1 --seam-cost-set-subme 4 --seam-cost-sad
  done
done
