#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2q
export ME_AB_MODES=t
for v in 3 7; do X265HIP_ME_CAND_VARIANT=$v timeout 200 python tools/me_ab_probe.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2q/nt.txt
for r in 24 32; do R=$r ME_AB_BOTH=1 timeout 200 python tools/me_ab_probe.py 2>&1 | grep -v amdgpu.ids; done >> gpurun_out/r2q/nt.txt
W=1920 H=1080 ME_AB_BOTH=1 timeout 200 python tools/me_ab_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2q/nt.txt
cat gpurun_out/r2q/nt.txt
