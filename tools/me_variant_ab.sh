#!/bin/bash
# A/B of the record-per-lane search kernel's switches under the current default surface format (measurement aid):
# X265HIP_ME_CAND_VARIANT bit 0 = source CTU in LDS, bit 1 = odd dword pairs shuffled, bit 2 = nontemporal stores
for v in 3 7 3 7 2 1 0 6; do echo "== variant $v: $(X265HIP_ME_CAND_VARIANT=$v python bench.py --no-encoder --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stages_ms']['me'])")"; done
