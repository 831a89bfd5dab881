#!/bin/bash
# search -> refinement -> reconstruction in parts of CTU rows on two streams: same checksums, step time against the number of parts
cd "$GRAFT_REPO_ROOT"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]["checksum"]; print(d["value"], d["ms_per_step"], c["subpel"], c["dist"], c["recon"], c["recon_cb"], c["recon_cr"], c["sao_count"])'
for sp in 1 2 3 4 6 1 2; do
  echo "== split=$sp"
  timeout 200 python bench.py --steps 60 --warmup 5 --no-encoder --no-cpu-baseline --split $sp 2>gpurun_out/split_err.log | python -c "$show" || tail -5 gpurun_out/split_err.log
done
