#!/bin/bash
# round 2, GPU visit K: new tests (TU tables, error policy, phase planes + subpel seam), then the encoder bench with all three seams
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_phase_planes.py tests/test_gpu_seam.py tests/test_gpu_table.py -x -q -m gpu -k "phase or subpel or error_policy" > gpurun_out/r2k/new_tests.txt 2>&1
tail -5 gpurun_out/r2k/new_tests.txt
timeout 600 python -m pytest tests/test_gpu_recon.py tests/test_gpu_intra_recon.py -x -q -m gpu -k "scaling" > gpurun_out/r2k/tables_tests.txt 2>&1
tail -5 gpurun_out/r2k/tables_tests.txt
GPU_MAX_HW_QUEUES=16 timeout 900 python tools/encoder_bench.py --configs cfg3 --tables c,seam --frames 8 --seam-range 24 --seam-lookahead --seam-subpel > gpurun_out/r2k/enc_cfg3_all.json 2> gpurun_out/r2k/enc_cfg3_all.log
tail -3 gpurun_out/r2k/enc_cfg3_all.log
GPU_MAX_HW_QUEUES=16 timeout 600 python tools/encoder_bench.py --configs cfg3 --tables seam --frames 8 --seam-range 24 --seam-subpel > gpurun_out/r2k/enc_cfg3_sub.json 2> gpurun_out/r2k/enc_cfg3_sub.log
tail -2 gpurun_out/r2k/enc_cfg3_sub.log
