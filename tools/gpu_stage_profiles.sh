#!/bin/bash
# rocprofv3 kernel-trace summaries of the stage probes (search drivers, lookahead cost estimate, intra TU candidates, SAO passes).
# Usage (from the repo root on the GPU box):  bash tools/gpu_stage_profiles.sh [tag]   -> gpurun_out/<tag>/stage_kernel_stats.txt
set -u
TAG=${1:-stages}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
: > "$OUT/stage_kernel_stats.txt"
run() {   # name, command...
    local name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    echo "## $*" >> "$OUT/stage_kernel_stats.txt"
    python "$ROOT/tools/rocprof_summary.py" kernel-trace $(find "$OUT/$name" -name '*.db' | head -1) 2>&1 | grep -v "at::native\|rocclr\|^# rocprofv3" | head -12 >> "$OUT/stage_kernel_stats.txt"
    find "$OUT/$name" -name '*.db' -delete
}
run search python "$ROOT/bench.py" --search-probe --width 3840 --height 2160
run lookahead python "$ROOT/bench.py" --lookahead-probe --width 3840 --height 2160
run intratu python "$ROOT/bench.py" --prims --only intratu,sao --no-cpu
cat "$OUT/stage_kernel_stats.txt"
