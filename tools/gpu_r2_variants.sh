#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2v
run() { echo "# bench.py --no-cpu-baseline $*"; timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','unit','ms_per_step','stages_ms')}), json.dumps({k:d['roofline'][k] for k in ('kernel','achieved','frac','launch_ms')}))"; }
{ run --depth 10; run --width 1920 --height 1080; run --width 1920 --height 1080 --subme 2; run --width 7680 --height 4320 --depth 10 --steps 6 --warmup 2; run --search star; run --search hex; run --no-surface; run --surf-format packed; run --banded --band-rows 4; } > gpurun_out/r2v/variants.txt 2>&1
cat gpurun_out/r2v/variants.txt
