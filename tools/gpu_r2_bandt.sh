#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["checksum"])'
for rows in 2 3 4 5; do for t in 1; do
  echo "== banded ($rows CTU rows), record-per-lane kernel from $t rows"
  X265HIP_BAND_T_ROWS=$t timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows 2>gpurun_out/bt_err.log | python -c "$show" || tail -5 gpurun_out/bt_err.log
done; done
