#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2bp
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2bp/stats" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-encoder --banded --band-rows 4 > "$GRAFT_REPO_ROOT/gpurun_out/r2bp/bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2bp/err.txt"
cd "$GRAFT_REPO_ROOT"
python tools/rocprof_summary.py kernel-trace $(find gpurun_out/r2bp/stats -name '*.db' | head -1) > gpurun_out/r2bp/kernel_stats.txt 2>&1
head -32 gpurun_out/r2bp/kernel_stats.txt
find gpurun_out/r2bp -name '*.db' -delete
