#!/bin/bash
# round 2, GPU visit L: the candidate-per-lane exhaustive search (me_cand_kernel.hip): parity, then A/B timing against the row-walking kernel
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2l
timeout 600 python -m pytest tests/test_gpu_me.py -x -q -m gpu > gpurun_out/r2l/me_tests.txt 2>&1
tail -15 gpurun_out/r2l/me_tests.txt
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench_cand.json 2> gpurun_out/r2l/bench_cand.log
tail -c 1500 gpurun_out/r2l/bench_cand.json
X265HIP_ME_KERNEL=rows timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench_rows.json 2> gpurun_out/r2l/bench_rows.log
tail -c 600 gpurun_out/r2l/bench_rows.json
