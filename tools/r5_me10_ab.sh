#!/bin/bash
# Round 5: the 16-bit minima-only search launch, round 4's kernel (X265HIP_ME_W2=0) against me_ctu_w2_kernel (=1) - one box, interleaved, three rounds; 4K and 8K 10-bit
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --depth 10 --steps 40 --warmup 4 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W2', os.environ.get('X265HIP_ME_W2'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'], d['roofline']['kernel'])" "$@"; }
for round in 1 2 3; do
  for v in 0 1; do X265HIP_ME_W2=$v run; done
done
for v in 0 1; do X265HIP_ME_W2=$v run --width 7680 --height 4320 --steps 8 --warmup 2; done
