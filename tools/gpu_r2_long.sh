#!/bin/bash
# longer clips: the 25-picture lookahead of preset slow is in steady state, not flushed once
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2long
export GPU_MAX_HW_QUEUES=16
timeout 900 python tools/encoder_bench.py --configs cfg3 --tables c,seam --frames 48 --frame-threads 5 --seam-range 24 --seam-lookahead > gpurun_out/r2long/F5.json 2> gpurun_out/r2long/F5.log
grep "^\[enc" gpurun_out/r2long/F5.log | cut -c1-330
timeout 900 python tools/encoder_bench.py --configs cfg3 --tables c,seam --frames 48 --frame-threads 1 --seam-range 24 --seam-lookahead --seam-subpel > gpurun_out/r2long/F1.json 2> gpurun_out/r2long/F1.log
grep "^\[enc" gpurun_out/r2long/F1.log | cut -c1-330
