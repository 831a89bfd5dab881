#!/bin/bash
# functional dry run of the N > 1 ring on ONE GPU (ranks share the device, bands travel through host memory: gloo) - the band streams,
# the per-band transfer ordering and the automatic band size on real kernels; the checksums of every N must agree with the 1-GPU banded run
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2dry
export X265HIP_BENCH_BACKEND=gloo
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) bench.py --gpus $n --steps 4 --warmup 2 > gpurun_out/r2dry/ring$n.json 2> gpurun_out/r2dry/ring$n.err
  echo "ring N=$n rc=$?"; tail -3 gpurun_out/r2dry/ring$n.err | cut -c1-300
  python - <<PY
import json
d=json.loads(open('gpurun_out/r2dry/ring$n.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d['config']['parallelism'][:150], d['config']['checksum'])
PY
done
