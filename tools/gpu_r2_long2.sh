#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2long
export GPU_MAX_HW_QUEUES=16
timeout 1200 python tools/encoder_bench.py --configs cfg4 --tables c,seam --frames 32 --frame-threads 5 --seam-range 24 --seam-lookahead > gpurun_out/r2long/cfg4_F5.json 2> gpurun_out/r2long/cfg4_F5.log
grep "^\[enc" gpurun_out/r2long/cfg4_F5.log | cut -c1-260
timeout 600 python tools/encoder_bench.py --configs cfg2 --tables c,seam --frames 96 --frame-threads 3 --seam-range 24 --seam-lookahead > gpurun_out/r2long/cfg2_F3.json 2> gpurun_out/r2long/cfg2_F3.log
grep "^\[enc" gpurun_out/r2long/cfg2_F3.log | cut -c1-260
