#!/bin/bash
# Round 5: why does reading the window as two 64-bit values per row lose?  LDS counters of the minima-only search launch for the default flag set (254: three dwords per row),
# LDA (446: two aligned ds_read_b64 from two window copies) and LD64 (1: 4-byte-aligned 64-bit loads) - one pass per counter group, --kernel-trace only.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5lds}; mkdir -p "$OUT"; export TMPDIR=/tmp
BASE="--no-cpu-baseline --no-encoder --no-verify --steps 6 --warmup 2"
for fl in 254 446 1; do
  i=0
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1))
    ( cd /tmp && X265HIP_ME_Q2_FLAGS=$fl timeout 600 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/lds_${fl}_$i" -o b -- python "$ROOT/bench.py" $BASE > /dev/null 2> "$OUT/lds_${fl}_$i.err" )
  done
  python tools/rocprof_summary.py pmc $(find "$OUT" -path "*lds_${fl}_*" -name '*.db') > "$OUT/lds_$fl.txt" 2>&1
  find "$OUT" -name '*.db' -delete
  echo "=== flags $fl"; grep "me_ctu_q2" "$OUT/lds_$fl.txt" | cut -c1-40,92-200
done
