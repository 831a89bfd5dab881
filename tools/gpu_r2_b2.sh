#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2z
timeout 900 python bench.py > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err; echo "bench rc=$?"
grep "^\[encoder" gpurun_out/r2z/bench.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2z/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','bit_exact')}); print(d['encoder_summary'])
PY
