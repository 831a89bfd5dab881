#!/bin/bash
# round 5, GPU visit f: the LDA flag (two window copies), wavefronts per workgroup for the new kernels, the randomised soak, the slow table test's per-call time
OUT=gpurun_out/r5f; mkdir -p $OUT
python -m pytest tests/test_gpu_me.py -m gpu -q -k "minima_only" > $OUT/me_variants.log 2>&1; tail -3 $OUT/me_variants.log
R5_ME_FLAGS="254 446 256" bash tools/r5_me_ab.sh > $OUT/me_ab_lda.txt 2>&1; cat $OUT/me_ab_lda.txt
run() { timeout 300 python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves', os.environ.get('X265HIP_ME_BEST_WAVES'), sys.argv[1:], 'step', d['ms_per_step'], 'me', d['stages_ms']['me'])" "$@"; }
for r in 1 2; do for w in 8 12 16; do X265HIP_ME_BEST_WAVES=$w run; done; done > $OUT/me_waves.txt 2>&1
for r in 1 2; do for w in 8 12 16; do X265HIP_ME_BEST_WAVES=$w run --depth 10; done; done >> $OUT/me_waves.txt 2>&1; cat $OUT/me_waves.txt
python tools/r5_me_minima_soak.py --seconds 75 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
( time python -m pytest tests/test_ref_encoder.py -m gpu -q -s -k full_size ) > $OUT/table_full_size.log 2>&1; grep -E "T3 full size|passed|failed|^real" $OUT/table_full_size.log
