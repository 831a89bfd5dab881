#!/bin/bash
# ONE parametrised GPU-box visit (replaces the round-2 tools/gpu_r2_*.sh one-offs).  Run from the repo root on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/gpu_visit.sh <tag> <step> [<step> ...]'
# Steps (each writes under gpurun_out/<tag>/ and prints a short tail):
#   tests[:<pytest args>]   python -m pytest tests -m gpu -q [args]           -> pytest_N.log
#   smoke                   __graft_entry__.smoke()                             -> smoke.log
#   bench[:<args>]          python bench.py [args]                             -> bench[_N].json / .err
#   stats[:<args>]          rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-encoder [args]
#                                                                                -> kernel_stats[_N].txt
#   timeline[:<args>]       rocprofv3 --kernel-trace of the same command -> the dispatches of ONE steady-state step with start / end / idle gaps -> timeline_N.txt
#   pmc[:<args>]            two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of the same command -> pmc[_N].txt
#   variants                the bench variants table (10-bit, 1080p, 8K 10-bit, star / hex, no surfaces, banded) -> variants.txt
#   py:<script> [args]      python <script> args (':'-separated), e.g. py:tools/phase_probe.py:--depth:10      -> py_N.log
#   sh:<command>            bash -c '<command>' (rest of the step after sh:)                                  -> sh_N.log
set -u
TAG=${1:-visit}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
# the reference build under oracle/_ref travels with the snapshot (git-ignored, not gpurun-ignored): tests that need it FAIL instead of skipping when it is lost
export X265HIP_EXPECT_REF=${X265HIP_EXPECT_REF:-1}
n=0
BENCH_PROF="--steps 10 --warmup 2 --no-cpu-baseline --no-encoder"
for step in "$@"; do
    n=$((n + 1))
    kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
    case $kind in tests|bench|stats|pmc|timeline) arg=${arg//:/ } ;; esac          # ':' separates arguments (bench:--no-encoder:--steps:20)
    echo "=== [$n] $step"
    case $kind in
    tests)
        case "$arg" in *tests/*) where="" ;; *) where="tests" ;; esac          # explicit test files replace the whole-suite default (use -k:<one_word> for filters: ':' splits arguments)
        ( time timeout 1700 python -m pytest $where -m gpu -q --durations=6 $arg ) > "$OUT/pytest_$n.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_$n.log"
        grep -E "^(FAILED|ERROR)|passed|failed|^real" "$OUT/pytest_$n.log" | tail -30 ;;
    smoke)
        timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log" ;;
    bench)
        timeout 1200 python bench.py $arg > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; echo "bench rc=$?"
        python - "$OUT/bench_$n.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "ms_per_step", "bit_exact", "stages_ms", "roofline", "stages_roofline", "cpu_baseline", "encoder_summary")}
    print(json.dumps(keep)[:3000])
except Exception as e:
    print("bench line unreadable:", e)
PY
        tail -3 "$OUT/bench_$n.err" ;;
    stats)
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$n" -o bench -- python "$ROOT/bench.py" $BENCH_PROF $arg > "$OUT/bench_under_rocprof_$n.json" 2> "$OUT/stats_$n.err" )
        python tools/rocprof_summary.py kernel-trace $(find "$OUT/stats_$n" -name '*.db' | head -1) > "$OUT/kernel_stats_$n.txt" 2>&1 || true
        find "$OUT/stats_$n" -name '*.db' -delete
        grep -v "at::native\|rocclr" "$OUT/kernel_stats_$n.txt" | head -40 ;;
    timeline)
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$OUT/tl_$n" -o bench -- python "$ROOT/bench.py" $BENCH_PROF $arg > /dev/null 2> "$OUT/tl_$n.err" )
        python tools/rocprof_summary.py timeline $(find "$OUT/tl_$n" -name '*.db' | head -1) > "$OUT/timeline_$n.txt" 2>&1 || true
        find "$OUT/tl_$n" -name '*.db' -delete
        cat "$OUT/timeline_$n.txt" ;;
    pmc)
        for c in FETCH_SIZE WRITE_SIZE; do
            ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_$n" -o bench -- python "$ROOT/bench.py" $BENCH_PROF $arg > /dev/null 2> "$OUT/pmc_${c}_$n.err" )
        done
        python tools/rocprof_summary.py pmc $(find "$OUT/pmc_FETCH_SIZE_$n" -name '*.db' | head -1) $(find "$OUT/pmc_WRITE_SIZE_$n" -name '*.db' | head -1) > "$OUT/pmc_$n.txt" 2>&1 || true
        find "$OUT" -name '*.db' -delete
        grep -v "at::native\|rocclr" "$OUT/pmc_$n.txt" | cut -c1-60,92-200 | head -70 ;;
    variants)
        run() { echo "# bench.py --no-cpu-baseline --no-encoder $*"; timeout 300 python bench.py --no-cpu-baseline --no-encoder --steps 50 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','unit','ms_per_step','stages_ms')}), json.dumps({k:d['roofline'][k] for k in ('kernel','achieved','frac','launch_ms')}))"; }
        { run --depth 10; run --width 1920 --height 1080; run --width 7680 --height 4320 --depth 10 --steps 6 --warmup 2; run --search star; run --search hex; run --no-surface; run --banded --band-rows 4; } > "$OUT/variants.txt" 2>&1
        cat "$OUT/variants.txt" ;;
    py)
        timeout 1200 python ${arg//:/ } > "$OUT/py_$n.log" 2>&1; echo "rc=$?"; tail -40 "$OUT/py_$n.log" ;;
    sh)
        timeout 1500 bash -c "$arg" > "$OUT/sh_$n.log" 2>&1; echo "rc=$?"; tail -40 "$OUT/sh_$n.log" ;;
    *) echo "unknown step $step" ;;
    esac
done
ls -la "$OUT" | tail -30
