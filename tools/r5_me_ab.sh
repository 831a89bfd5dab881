#!/bin/bash
# Round 5: A/B of the 8-bit minima-only search launch ON ONE BOX, interleaved, three rounds (boxes differ by up to 6 %).  X265HIP_ME_Q2_FLAGS selects a
# flag set of me_ctu_q2_kernel (csrc/me_kernels.hip: 1 LD64, 2 CTAB, 4 PAIR64, 8 DEFERX, 16 COLMIN, 32 MASK, 64 RING, 128 QUAD64; 0 = round 4's instruction stream in the new
# kernel), "r4" = round 4's me_ctu_q_kernel<best> itself.
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags', os.environ.get('X265HIP_ME_Q2_FLAGS', 'r4'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
FLAGS=${R5_ME_FLAGS:-"0 62 64 126 190 254 238"}      # visit r5f: R5_ME_FLAGS="254 446 256" + X265HIP_ME_BEST_WAVES sweeps
for round in 1 2 3; do
  X265HIP_ME_BEST_VARIANT=0 run
  for f in $FLAGS; do X265HIP_ME_Q2_FLAGS=$f run; done
done
