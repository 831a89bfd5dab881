#!/bin/bash
# Round-6 profile visit (the round-4 recipe on round 6's library, every configuration again - round-5 verdict, weak 12 / next 5: the 10-bit twin and 8K need files of their own): for every BASELINE configuration bench.py can run on one GPU (4K 8-bit = configs[2], 4K 10-bit = configs[3], 8K 10-bit =
# configs[4]) the kernel trace, the HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only), the matrix-core occupancy
# (SQ_VALU_MFMA_BUSY_CYCLES) and - for the 16-bit search kernel - the SQ instruction counters; plus the --surface variant of the 4K 8-bit step
# (rounds 1 - 3's default) and the corrected VALU-rate microbenchmark.  Summaries land in gpurun_out/<tag>/; tools/r4_assemble_profiles.py turns
# them into profiles/r04_* and profiles/stage_traffic.json / traffic.json / valu_rates.json.
#   gpurun --timeout 2400 -- 'bash tools/r6_profile.sh r6p'; then python tools/r4_assemble_profiles.py gpurun_out/r6p r06
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-r6p}
mkdir -p "$OUT"
export TMPDIR=/tmp
BASE="--no-cpu-baseline --no-encoder --no-verify"
prof() {          # key, bench args...
    local key=$1; shift
    echo "=== $key: $*"
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$key" -o b -- python "$ROOT/bench.py" $BASE "$@" > "$OUT/bench_$key.json" 2> "$OUT/kt_$key.err" )
    python tools/rocprof_summary.py kernel-trace $(find "$OUT/kt_$key" -name '*.db' | head -1) > "$OUT/kernel_stats_$key.txt" 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_$key" -o b -- python "$ROOT/bench.py" $BASE "$@" > /dev/null 2> "$OUT/pmc_${c}_$key.err" )
    done
    python tools/rocprof_summary.py pmc $(find "$OUT/pmc_FETCH_SIZE_$key" -name '*.db' | head -1) $(find "$OUT/pmc_WRITE_SIZE_$key" -name '*.db' | head -1) > "$OUT/pmc_$key.txt" 2>&1
    ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d "$OUT/mfma_$key" -o b -- python "$ROOT/bench.py" $BASE "$@" > /dev/null 2> "$OUT/mfma_$key.err" )
    python tools/rocprof_summary.py pmc $(find "$OUT/mfma_$key" -name '*.db' | head -1) > "$OUT/mfma_$key.txt" 2>&1
    find "$OUT" -name '*.db' -delete
    grep -v "at::native\|rocclr" "$OUT/kernel_stats_$key.txt" | head -14
    tail -1 "$OUT/bench_$key.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline'].get('valu'))" 2>/dev/null
}
inst() {          # key, bench args...: SQ instruction counters, one pass per group
    local key=$1; shift
    local i=0
    for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
        i=$((i + 1))
        ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/inst_${key}_$i" -o b -- python "$ROOT/bench.py" $BASE --steps 6 --warmup 2 "$@" > /dev/null 2> "$OUT/inst_${key}_$i.err" )
    done
    python tools/rocprof_summary.py pmc $(find "$OUT" -path "*inst_${key}_*" -name '*.db') > "$OUT/inst_$key.txt" 2>&1
    find "$OUT" -name '*.db' -delete
    grep "me_ctu" "$OUT/inst_$key.txt" | cut -c1-60,92-200 | head -20
}
prof 3840x2160_d8 --steps 10 --warmup 2
prof 3840x2160_d10 --steps 10 --warmup 2 --depth 10
prof 7680x4320_d10 --steps 4 --warmup 2 --depth 10 --width 7680 --height 4320
inst 3840x2160_d10 --depth 10
inst 3840x2160_d8
python tools/fetch_calibration.py "$OUT" 2>&1 | tail -12
ls "$OUT" | head -60
