#!/bin/bash
# last visit of round 2: the whole GPU suite and the smoke call on the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2v
( time timeout 1200 python -m pytest tests -m gpu -q --durations=6 ) > gpurun_out/r2v/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r2v/pytest.log | tail -10
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1
