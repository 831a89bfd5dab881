#!/bin/bash
# Round 6 A/B (one box, interleaved): the reference's phase planes on the side stream NEXT TO the search (X265HIP_PREP_OVERLAP=1) against behind it (default) - round 3 measured no gain; again on the closing library
# result (visit r9f): 1.6105 / 1.6109 / 1.6074 ms behind the search against 1.6180 / 1.6168 / 1.6160 next to it - the default stays
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for r in 1 2 3; do
  for e in "" "X265HIP_PREP_OVERLAP=1"; do
    env $e python bench.py --no-encoder --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${e:-planes behind the search}', d['ms_per_step'], d.get('bit_exact'), flush=True)"
  done
done
