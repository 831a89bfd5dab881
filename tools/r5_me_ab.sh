#!/bin/bash
# Round 5: A/B of the 8-bit minima-only search launch ON ONE BOX, interleaved, three rounds (boxes differ by up to 6 %):
# X265HIP_ME_BEST_VARIANT 0 = round 4's kernel (me_ctu_q_kernel<best>), 4 / unset = round 5's me_ctu_q2_kernel (row constants in SGPRs from an LDS table,
# aligned 64-bit window loads, costX once per group, the 64x64 level reduced for two rows at a time), 5 = the same with per-column 8x8 minima
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', os.environ.get('X265HIP_ME_BEST_VARIANT'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
for round in 1 2 3; do
  for v in 0 4 5; do X265HIP_ME_BEST_VARIANT=$v run; done
done
for v in 0 4 5; do X265HIP_ME_BEST_VARIANT=$v run --width 1920 --height 1080; done
