#!/bin/bash
# Round 6 A/B (one box): the lookahead's per-thread streams with the device's highest priority (X265HIP_LA_PRIORITY=1) against plain streams (the default), interleaved.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for e in "" "X265HIP_LA_PRIORITY=1" "" "X265HIP_LA_PRIORITY=1" "" "X265HIP_LA_PRIORITY=1"; do
  QUEUES=16 EXTRA_ENV="$e" bash tools/r6_lookahead_queues_ab.sh
done
