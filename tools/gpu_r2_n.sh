#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_me.py tests/test_gpu_seam.py -x -q -m gpu > gpurun_out/r2n/tests.txt 2>&1
tail -6 gpurun_out/r2n/tests.txt
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2n/bench_t.json 2> gpurun_out/r2n/bench_t.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2n/bench_t.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stages_ms'], d['roofline']['kernel'], d['roofline']['frac'])
PY
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --surf-format packed > gpurun_out/r2n/bench_p.json 2> gpurun_out/r2n/bench_p.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2n/bench_p.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stages_ms'], d['roofline']['kernel'], d['roofline']['frac'])
PY
