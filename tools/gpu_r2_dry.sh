#!/bin/bash
# functional dry run of bench.py's N > 1 path on the 1-GPU box: 2 and 3 ranks share the GPU, bands travel through host memory (gloo)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2dry
export X265HIP_BENCH_BACKEND=gloo
for n in 2 3; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 6 --warmup 2 > gpurun_out/r2dry/n$n.json 2> gpurun_out/r2dry/n$n.err
  echo "N=$n rc=$?"; tail -c 900 gpurun_out/r2dry/n$n.json; echo; grep -v "amdgpu.ids\|^$\|\*\*\*\*\|OMP_NUM_THREADS" gpurun_out/r2dry/n$n.err | tail -5
done
