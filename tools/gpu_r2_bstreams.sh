#!/bin/bash
# bands alternating between HIP streams: parity, then the banded step for 1 / 2 / 3 streams and band sizes
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["checksum"])'
echo "== whole frame"; for i in 1 2; do timeout 200 python bench.py --steps 60 --warmup 5 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"; done
for rows in 4 2; do for st in 1 2 3; do for fuse in 1 0; do  [ $rows = 2 ] && [ $st = 1 ] && continue
  echo "== banded ($rows CTU rows), streams=$st X265HIP_FUSE_SAO=$fuse"
  X265HIP_FUSE_SAO=$fuse timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows --band-streams $st 2>gpurun_out/bs_err.log | python -c "$show" || tail -5 gpurun_out/bs_err.log
done; done; done
