#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2p
export ME_AB_MODES=t X265HIP_ME_KERNEL=cand
for v in 0 1 2 3; do X265HIP_ME_CAND_VARIANT=$v timeout 200 python tools/me_ab_probe.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2p/variants.txt
cat gpurun_out/r2p/variants.txt
X265HIP_ME_CAND_VARIANT=1 timeout 300 python -m pytest tests/test_gpu_me.py -x -q -m gpu 2>&1 | tail -2
X265HIP_ME_CAND_VARIANT=2 timeout 300 python -m pytest tests/test_gpu_me.py -x -q -m gpu 2>&1 | tail -2
