#!/bin/bash
# Round 5: the banded step (what a rank of the frame-parallel ring runs) with a CTU's column groups dealt over several workgroups when a launch has few CTUs
# (X265HIP_ME_SPLIT_GROUPS=1, the default) against one workgroup per CTU (=0) - one box, interleaved; ms per picture by band height
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --banded "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', os.environ.get('X265HIP_ME_SPLIT_GROUPS'), sys.argv[1:], d['ms_per_step'])" "$@"; }
for n in 1 2 3 4 6; do for v in 0 1; do X265HIP_ME_SPLIT_GROUPS=$v run --band-rows $n --steps 30 --warmup 4; done; done
for n in 2 3 4; do for v in 0 1; do X265HIP_ME_SPLIT_GROUPS=$v run --depth 10 --band-rows $n --steps 20 --warmup 4; done; done
for n in 1 2 3; do for v in 0 1; do X265HIP_ME_SPLIT_GROUPS=$v run --width 1920 --height 1080 --band-rows $n --steps 40 --warmup 4; done; done
for v in 0 1; do X265HIP_ME_SPLIT_GROUPS=$v run --width 7680 --height 4320 --depth 10 --band-rows 4 --steps 5 --warmup 2; done
