#!/bin/bash
for k in 0 8 9 10 12 11 15; do echo "X265HIP_SAO_RDO_DEBUG=$k"; X265HIP_SAO_RDO_DEBUG=$k python tools/sao_rdo_probe.py 2>&1 | grep "4K\|8K"; done
