#!/bin/bash
# fused SAO (x265hip_sao_planes) : parity, then A/B of the whole-frame and the banded step
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_sao.py tests/test_gpu_pipeline.py tests/test_gpu_banded.py -x -q -m gpu 2>&1 | tail -3
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("bit_exact"), d.get("stages_ms"))'
for fuse in 0 1 0 1; do
  echo "== whole frame, X265HIP_FUSE_SAO=$fuse"
  X265HIP_FUSE_SAO=$fuse timeout 200 python bench.py --steps 60 --warmup 5 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "$show"
done
for fuse in 0 1; do
  echo "== banded (4 CTU rows), X265HIP_FUSE_SAO=$fuse"
  X265HIP_FUSE_SAO=$fuse timeout 200 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows 4 2>/dev/null | python -c "$show"
done
