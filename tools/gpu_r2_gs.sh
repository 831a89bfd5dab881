#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["checksum"])'
for rows in 4 2; do for g in 0 1; do
  echo "== banded ($rows CTU rows), 3 streams, graphs=$g"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows --band-streams 3 --band-graphs $g 2>gpurun_out/gs_err.log | python -c "$show" || tail -5 gpurun_out/gs_err.log
done; done
