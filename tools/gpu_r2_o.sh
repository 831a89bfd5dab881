#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2o
export ME_AB_MODES=t
timeout 200 python tools/me_ab_probe.py > gpurun_out/r2o/ab_t.txt 2>&1
cat gpurun_out/r2o/ab_t.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WRREQ[A-Z0-9_a-z]*\|TCC_REQ[_a-z]*\|TCC_WRITE[_a-z]*\|TA_BUSY[_a-z]*\|TCP_TCC_WRITE_REQ[_a-z]*\|SQ_INSTS_VMEM_WR\|SQ_WAIT_ANY\|TCC_EA0_WRREQ_STALL[_a-z]*\|TA_TA_BUSY[_a-z]*\|TCC_BUSY[_a-z]*\|TCP_PENDING_STALL_CYCLES[_a-z]*\|SQ_INST_CYCLES_VMEM[_A-Za-z]*\|TCC_TAG_STALL[_a-z]*\|TCC_EA0_WR_UNCACHED_32B[_a-z]*" | sort -u > "$GRAFT_REPO_ROOT/gpurun_out/r2o/avail.txt"
cat "$GRAFT_REPO_ROOT/gpurun_out/r2o/avail.txt" | tr '\n' ' '
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_STALL_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "TA_BUSY_avr TCC_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  ME_AB_BOTH=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/gpurun_out/r2o/$n" -o p --output-format csv -- python "$GRAFT_REPO_ROOT/tools/me_ab_probe.py" > "$GRAFT_REPO_ROOT/gpurun_out/r2o/$n.log" 2>&1
  tail -2 "$GRAFT_REPO_ROOT/gpurun_out/r2o/$n.log"
done
ls -R "$GRAFT_REPO_ROOT/gpurun_out/r2o" | head -30
