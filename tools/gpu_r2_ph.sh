#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_phase_planes.py tests/test_gpu_subpel.py tests/test_gpu_seam.py -x -q -m gpu -k "phase or subpel" 2>&1 | tail -2
timeout 100 python tools/phase_probe.py 2>&1 | grep phase_planes
timeout 200 python bench.py --steps 60 --warmup 5 --no-encoder --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
