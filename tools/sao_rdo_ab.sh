#!/bin/bash
# Timing breakdown of the serial SAO pass (measurement aid): X265HIP_SAO_RDO_DEBUG bits 1 / 2 / 4 switch the merge lanes / the copy wavefronts /
# the decision lanes off (results are then wrong - timing only); X265HIP_SAO_RDO_KERNEL=1 is the first organisation.
for d in ${SAO_AB_SET:-0 1 2 4 3 5 6 7}; do
    echo "== rows2, debug $d"; X265HIP_SAO_RDO_DEBUG=$d python tools/sao_rdo_probe.py 2>&1 | grep -v amdgpu.ids
done
if [ -z "${SAO_AB_SET:-}" ]; then echo "== first organisation"; X265HIP_SAO_RDO_KERNEL=1 python tools/sao_rdo_probe.py 2>&1 | grep -v amdgpu.ids; fi
