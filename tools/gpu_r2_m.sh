#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2m
{ timeout 200 python tools/me_ab_probe.py; X265HIP_ME_KERNEL=rows timeout 200 python tools/me_ab_probe.py; } > gpurun_out/r2m/ab.txt 2>&1
cat gpurun_out/r2m/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU -d "$GRAFT_REPO_ROOT/gpurun_out/r2m/pmc1" -o pmc1 --output-format csv -- python "$GRAFT_REPO_ROOT/tools/me_ab_probe.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2m/pmc1.log" 2>&1
tail -3 "$GRAFT_REPO_ROOT/gpurun_out/r2m/pmc1.log"
ls -R "$GRAFT_REPO_ROOT/gpurun_out/r2m" | head -20
