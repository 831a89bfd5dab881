#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { timeout 200 python bench.py --steps 100 --warmup 5 --no-encoder --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo "parallel=0"; run --parallel-planes 0
echo "parallel=1 prep overlapped with the search"; X265HIP_PREP_OVERLAP=1 run --parallel-planes 1
echo "parallel=1 prep after the search"; X265HIP_PREP_OVERLAP=0 run --parallel-planes 1
done
