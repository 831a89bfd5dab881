#!/bin/bash
# the 8-GPU configurations of BASELINE.json are 10-bit: whole picture against bands on one GPU
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["stages_ms"].get("me"))'
echo "== 4K 10-bit whole picture"; timeout 300 python bench.py --depth 10 --steps 40 --warmup 3 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"
for rows in 2 4; do echo "== 4K 10-bit bands of $rows rows"; timeout 300 python bench.py --depth 10 --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows 2>/dev/null | python -c "$show"; done
echo "== 8K 10-bit whole picture"; timeout 300 python bench.py --depth 10 --width 7680 --height 4320 --steps 8 --warmup 2 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"
for rows in 2 4; do echo "== 8K 10-bit bands of $rows rows"; timeout 300 python bench.py --depth 10 --width 7680 --height 4320 --steps 8 --warmup 2 --no-encoder --no-cpu-baseline --banded --band-rows $rows 2>/dev/null | python -c "$show"; done
