#!/bin/bash
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags', os.environ.get('X265HIP_ME_Q2_FLAGS'), 'waves', os.environ.get('X265HIP_ME_BEST_WAVES'), 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
for r in 1 2; do
  X265HIP_ME_Q2_FLAGS=254 run
  for w in 12 16; do X265HIP_ME_Q2_FLAGS=446 X265HIP_ME_BEST_WAVES=$w run; done
done
