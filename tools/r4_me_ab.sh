#!/bin/bash
# A/B of the minima-only search launch's variants ON ONE BOX (boxes differ by up to 6 %: comparisons across visits are worthless), interleaved, three rounds:
# X265HIP_ME_BEST_VARIANT bit 0 = per-column running minima for the 8x8 level, bit 1 = no scheduling fence between rows
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', os.environ.get('X265HIP_ME_BEST_VARIANT'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
for round in 1 2 3; do
  for v in 0 1 2 3; do X265HIP_ME_BEST_VARIANT=$v run; done
  for v in 0 1 2 3; do X265HIP_ME_BEST_VARIANT=$v run --depth 10; done
done
