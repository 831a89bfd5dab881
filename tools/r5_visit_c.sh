#!/bin/bash
# round 5, GPU visit c: ME flag A/B, the configurations verified in flight, two service instances, the seam matrix
OUT=gpurun_out/r5c; mkdir -p $OUT
python -m pytest tests/test_gpu_me.py -m gpu -q -k minima_only > $OUT/me_variants.log 2>&1; tail -3 $OUT/me_variants.log
bash tools/r5_me_ab.sh > $OUT/me_ab.txt 2>&1; cat $OUT/me_ab.txt
( time python -m pytest tests/test_gpu_seam.py -m gpu -q -s --durations=8 -k "verified_in_flight or saturated or two_service" ) > $OUT/seam_verified.log 2>&1; grep -E "verified:|passed|failed|FAILED|ERROR|^real|s call" $OUT/seam_verified.log | tail -20
bash tools/r5_seam_matrix.sh 2 > $OUT/seam_matrix.txt 2>&1; cat $OUT/seam_matrix.txt
