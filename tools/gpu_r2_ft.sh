#!/bin/bash
# how does the reference scale with its own frame threads on this box, and does the lookahead seam hold at -F > 1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2ft
export GPU_MAX_HW_QUEUES=16
for F in 1 3 5; do
  timeout 600 python tools/encoder_bench.py --configs cfg3 --tables c --frames 16 --frame-threads $F > gpurun_out/r2ft/c_F$F.json 2> gpurun_out/r2ft/c_F$F.log
  grep "^\[enc" gpurun_out/r2ft/c_F$F.log | cut -c1-200
done
timeout 600 python tools/encoder_bench.py --configs cfg3 --tables c,seam --frames 16 --frame-threads 5 --seam-range 24 --seam-lookahead --seam-subpel > gpurun_out/r2ft/seam_F5.json 2> gpurun_out/r2ft/seam_F5.log
grep "^\[enc" gpurun_out/r2ft/seam_F5.log | cut -c1-2400
