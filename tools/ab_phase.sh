#!/bin/bash
for k in 1 2 0; do echo "X265HIP_PHASE_KERNEL=$k"; X265HIP_PHASE_KERNEL=$k python tools/phase_probe.py 2>&1 | grep luma; done
