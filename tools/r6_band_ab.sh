#!/bin/bash
# Round 6: what the banded step costs at the LARGE bands the mini-GOP ring wants (the par column of profiles/r06_band_ab.txt came from a hook since removed).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
B="--no-cpu-baseline --no-encoder --no-verify --banded --steps 30 --warmup 4"
for rows in 4 8 12 17; do
  for par in 0 1; do
    for st in 2 3; do
      v=$(python bench.py $B --band-rows $rows --band-streams $st 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "rows $rows par $par streams $st: $v ms"
    done
  done
done
python bench.py --no-cpu-baseline --no-encoder --no-verify --steps 30 --warmup 4 2>/dev/null | python -c "import json,sys; print('whole picture', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
