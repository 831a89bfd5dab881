#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2dry
timeout 300 python -m pytest tests/test_gpu_me.py -x -q -m gpu 2>&1 | tail -1
export X265HIP_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 6 --warmup 2 --sharding gop > gpurun_out/r2dry/gop2.json 2> gpurun_out/r2dry/gop2.err
echo "gop N=2 rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2dry/gop2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d['config']['parallelism'][:60])
PY
