#!/bin/bash
# Round 6: the mini-GOP ring on real kernels - the ring tests, then bench.py's own N > 1 path on ONE GPU shared by 2 and 3 ranks over gloo (a functional dry run: processes
# sharing a device are time-sliced, the numbers mean nothing), chain (--ring-gop 0) and mini-GOPs (default 5, and 2 so that anchors alternate between the two ranks).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/${1:-r6r}; mkdir -p $OUT
export X265HIP_EXPECT_REF=1
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_ring_abi.py tests/test_gpu_banded.py -m gpu -q 2>&1 | tail -3
run() { tag=$1; n=$2; shift 2
  X265HIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29517 + RANDOM % 200)) bench.py --gpus $n --steps 6 --warmup 2 --width 1920 --height 1080 --no-encoder "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "$tag rc=$?"; tail -1 $OUT/bench_$tag.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','n_gpus')}, d['config'].get('ring'), d['config'].get('band_rows'), d.get('replicas',{}).get('value'))"; }
run 2ranks_gop5 2
run 2ranks_gop2 2 --ring-gop 2
run 2ranks_chain 2 --ring-gop 0
run 3ranks_gop5 3
# round 6, second half: the one-communicator broadcast transport's torch.distributed twin over the same real stages (X265HIP_RING_TRANSPORT=bcast falls to dist_bcast on gloo)
X265HIP_RING_TRANSPORT=bcast run 3ranks_gop5_bcast 3
X265HIP_RING_TRANSPORT=bcast run 2ranks_gop2_bcast 2 --ring-gop 2
