#!/bin/bash
# wavefronts per workgroup of the 8-bit minima-only search launch, interleaved on one box (X265HIP_ME_BEST_WAVES)
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves', os.environ.get('X265HIP_ME_BEST_WAVES'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
for round in 1 2 3; do
  for w in 16 12 10 8; do X265HIP_ME_BEST_WAVES=$w run --depth 10; done
done
for w in 16 12 8; do X265HIP_ME_BEST_WAVES=$w run --depth 10 --width 7680 --height 4320 --steps 6; done
