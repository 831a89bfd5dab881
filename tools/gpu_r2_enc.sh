#!/bin/bash
# the two 10-bit BASELINE configurations at encoder level: C table vs all three seams
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2e
export GPU_MAX_HW_QUEUES=16
timeout 600 python tools/encoder_bench.py --configs cfg4 --tables c,seam --frames 4 --seam-range 24 --seam-lookahead --seam-subpel --seam-subpel-slots 6 > gpurun_out/r2e/cfg4.json 2> gpurun_out/r2e/cfg4.log
grep "^\[enc" gpurun_out/r2e/cfg4.log | cut -c1-1500
timeout 1500 python tools/encoder_bench.py --configs cfg5 --tables c,seam --frames 3 --seam-range 24 --seam-slots 6 --seam-lookahead --seam-subpel --seam-subpel-slots 5 > gpurun_out/r2e/cfg5.json 2> gpurun_out/r2e/cfg5.log
grep "^\[enc" gpurun_out/r2e/cfg5.log | cut -c1-1500; tail -3 gpurun_out/r2e/cfg5.log | cut -c1-300
