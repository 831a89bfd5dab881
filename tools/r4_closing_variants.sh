#!/bin/bash
# the closing library's step at the other BASELINE geometries / depths (the default line is 4K 8-bit): one line per variant
run() { echo "# bench.py --no-cpu-baseline --no-encoder $*"; timeout 300 python bench.py --no-cpu-baseline --no-encoder --steps 50 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(json.dumps({k:d[k] for k in ('value','unit','ms_per_step','bit_exact','stages_ms') if k in d}), json.dumps({'kernel': r['kernel'], 'frac': r['frac'], 'launch_ms': r['launch_ms'], 'valu_frac': (r.get('valu') or {}).get('frac')}))"; }
run; run --surface; run --depth 10; run --width 7680 --height 4320 --depth 10 --steps 8 --warmup 2; run --width 1920 --height 1080; run --search star; run --banded --band-rows 4
