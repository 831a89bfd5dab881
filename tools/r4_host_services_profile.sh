#!/bin/bash
# wall time per call + rocprofv3 kernel table of x265hip_weight_analyse_host / x265hip_aq_frame_host at 4K -> gpurun_out/<tag>/host_services*.txt
OUT=${1:-gpurun_out/r4z}; mkdir -p $OUT
export TMPDIR=/tmp
python tools/host_services_probe.py --depth 8 > $OUT/host_services.txt 2>&1
python tools/host_services_probe.py --depth 10 >> $OUT/host_services.txt 2>&1
R=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hsprof -o hs -- python $R/tools/host_services_probe.py --depth 8 --reps 3 > /tmp/hsprof.log 2>&1 )
echo "# tools/host_services_probe.py --depth 8 --reps 3 (4K: 2 weightAnalyse cases x 9 calls, 2 AQ cases x 4 calls)" > $OUT/host_services_kernel_stats.txt
python tools/rocprof_summary.py kernel-trace $(find /tmp/hsprof -name '*.db' | head -1) 2>&1 | grep -v "at::native\|rocclr" >> $OUT/host_services_kernel_stats.txt
cat $OUT/host_services.txt; head -20 $OUT/host_services_kernel_stats.txt
