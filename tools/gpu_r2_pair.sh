#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_recon.py tests/test_gpu_deblock.py tests/test_gpu_pipeline.py tests/test_gpu_banded.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("bit_exact"), d["config"]["checksum"])'
timeout 300 python bench.py --steps 100 --warmup 5 --no-encoder 2>/dev/null | python -c "$show"
for rows in 4 2; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-encoder --no-cpu-baseline --banded --band-rows $rows 2>/dev/null | python -c "$show"
done
