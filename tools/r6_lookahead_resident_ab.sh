#!/bin/bash
# Round 6 A/B (one box): the frame-cost estimates with (a) the searched vectors resident on the device + one pinned upload / download per call, (b) a split B estimate's two
# lists walked side by side - each against the form before it (X265HIP_LA_RESIDENT_OFF=1, X265HIP_LOWRES_COST_SO_OFF=1).  Per-kind call times by X265HIP_LA_STATS.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for e in "" "X265HIP_LA_RESIDENT_OFF=1" "X265HIP_LOWRES_COST_SO_OFF=1" "X265HIP_LA_RESIDENT_OFF=1 X265HIP_LOWRES_COST_SO_OFF=1" "" "X265HIP_LA_RESIDENT_OFF=1 X265HIP_LOWRES_COST_SO_OFF=1"; do
  QUEUES=16 EXTRA_ENV="$e" bash tools/r6_lookahead_queues_ab.sh
done
