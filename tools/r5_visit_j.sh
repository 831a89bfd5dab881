#!/bin/bash
OUT=gpurun_out/r5j; mkdir -p $OUT
python -m pytest tests/test_gpu_me.py -m gpu -q -k "minima_only and (q446 or q256 or q254)" > $OUT/me.log 2>&1; tail -2 $OUT/me.log
R5_ME_FLAGS="254 446" bash tools/r5_me_ab.sh > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
export TMPDIR=/tmp; ROOT=$(pwd)
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
( cd /tmp && X265HIP_ME_Q2_FLAGS=446 timeout 600 rocprofv3 --pmc $grp --kernel-trace -d "$ROOT/$OUT/c" -o b -- python "$ROOT/bench.py" --no-cpu-baseline --no-encoder --no-verify --steps 6 --warmup 2 > /dev/null 2> $ROOT/$OUT/c.err )
done
python tools/rocprof_summary.py pmc $(find $OUT/c -name '*.db') 2>/dev/null | grep me_ctu_q2 | cut -c1-40,92-200
find $OUT -name '*.db' -delete
