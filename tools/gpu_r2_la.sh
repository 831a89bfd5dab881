#!/bin/bash
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["checksum"]["intra_cost"], d["config"]["checksum"]["recon"])'
for v in 0 1 0 1; do
  echo "== X265HIP_LA_AFTER_ME=$v"
  X265HIP_LA_AFTER_ME=$v timeout 200 python bench.py --steps 100 --warmup 5 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"
done
