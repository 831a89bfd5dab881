#!/bin/bash
# banded step against the band size (three band streams, fused SAO)
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for rows in 1 2 3 4 6 8 12 17; do
  echo "== banded ($rows CTU rows), streams=3"
  timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows --band-streams 3 2>gpurun_out/bs_err.log | python -c "$show" || tail -5 gpurun_out/bs_err.log
done
echo "== banded (4 CTU rows), streams=4"
timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows 4 --band-streams 4 2>gpurun_out/bs_err.log | python -c "$show"
echo "== banded (2 CTU rows), streams=4"
timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows 2 --band-streams 4 2>gpurun_out/bs_err.log | python -c "$show"
