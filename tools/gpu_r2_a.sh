#!/bin/bash
# Round-2 GPU visit A: parity suite (incl. the full-size tests), default bench line with bit_exact, encoder-level legs (C table vs per-call HIP stubs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2a
mkdir -p "$OUT"
cd "$ROOT"
nproc > "$OUT/host.txt"; cat /sys/fs/cgroup/cpu.max >> "$OUT/host.txt" 2>&1; free -g >> "$OUT/host.txt"; cat /sys/fs/cgroup/memory.max >> "$OUT/host.txt" 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -25 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
timeout 500 python tools/encoder_bench.py --configs cfg1,cfg2 --tables c,hip --budget-s 60 > "$OUT/encoder_cfg12.json" 2> "$OUT/encoder_cfg12.err"; echo "enc12 rc=$?"; grep "^\[enc" "$OUT/encoder_cfg12.err"
timeout 600 python tools/encoder_bench.py --configs cfg3 --tables c,hip --frames 4 --budget-s 200 > "$OUT/encoder_cfg3.json" 2> "$OUT/encoder_cfg3.err"; echo "enc3 rc=$?"; grep "^\[enc" "$OUT/encoder_cfg3.err"
ls -la "$OUT"
