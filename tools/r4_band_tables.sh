#!/bin/bash
# banded step time per band height, per (bit depth, picture size), one GPU (round-3 verdict, next 8: band tables measured, not scaled)
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --banded "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['ms_per_step'])" "$@"; }
for n in 1 2 3 4 5 6 8 12 17; do run --band-rows $n --steps 30 --warmup 4; done
for n in 2 3 4 6 8 12 17; do run --depth 10 --band-rows $n --steps 20 --warmup 4; done
for n in 4 6 8 12 17 34; do run --depth 10 --width 7680 --height 4320 --band-rows $n --steps 5 --warmup 2; done
for n in 1 2 3 5 9; do run --width 1920 --height 1080 --band-rows $n --steps 40 --warmup 4; done
