#!/bin/bash
# HIP graphs per band: parity, then banded step with / without graphs and with / without the fused SAO
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_sao.py tests/test_gpu_banded.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"]["checksum"])'
for g in 0 1; do for fuse in 0 1; do
  echo "== banded (4 CTU rows), graphs=$g X265HIP_FUSE_SAO=$fuse"
  X265HIP_FUSE_SAO=$fuse timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows 4 --band-graphs $g 2>gpurun_out/graph_err_$g$fuse.log | python -c "$show" || tail -5 gpurun_out/graph_err_$g$fuse.log
done; done
echo "== whole frame"
timeout 200 python bench.py --steps 60 --warmup 5 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"
