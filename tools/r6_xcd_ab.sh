#!/bin/bash
# Round 6 A/B (one box, interleaved): the search kernels' XCD-aware workgroup -> CTU order against raster order (X265HIP_ME_XCD_OFF=1); bench step + the search launch by HIP events
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for r in 1 2 3; do
  for e in "" "X265HIP_ME_XCD_OFF=1"; do
    env $e python bench.py --no-encoder --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${e:-xcd order}', d['ms_per_step'], 'me', d['stages_ms']['me'], 'launch', d['roofline']['launch_ms'], flush=True)"
  done
done
