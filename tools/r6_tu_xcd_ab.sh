#!/bin/bash
# Round 6 A/B (one box, interleaved): the persistent TU kernels' XCD-aware block order against raster order, luma and chroma separately (X265HIP_TU_XCD_OFF=1 / luma / chroma)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for r in 1 2 3; do
  for e in "" "X265HIP_TU_XCD_OFF=chroma" "X265HIP_TU_XCD_OFF=luma" "X265HIP_TU_XCD_OFF=1"; do
    env $e python bench.py --no-encoder --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${e:-xcd order}', d['ms_per_step'], 'recon', d['stages_ms']['recon'], 'recon_chroma', d['stages_ms']['recon_chroma'], flush=True)"
  done
done
