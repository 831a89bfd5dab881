"""Golden vectors: reference outputs recorded from the real reference build (tools/gen_golden.py,
committed under tests/golden/) replayed on the oracle table (CPU) and on the HIP table (GPU)."""
import os

import numpy as np
import pytest

import golden_cases as G
import harness as H


def _store(depth, root):
    path = os.path.join(root, "tests", "golden", f"prims_d{depth}.npz")
    assert os.path.exists(path), "golden fixtures missing: run tools/gen_golden.py where /root/reference exists"
    return dict(np.load(path))


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_reproduces_golden_vectors(depth, repo_root):
    store = _store(depth, repo_root)
    orc = H.load_oracle(depth, repo_root)
    fails = []
    cases = G.cases(depth)
    for case in cases:
        fails += G.run_case(orc, case, store, record=False)
    assert len(cases) > 400
    assert not fails, "\n".join(fails[:30])


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 10])
def test_hip_table_reproduces_golden_vectors(depth, repo_root):
    from test_gpu_table import load_hip_table
    store = _store(depth, repo_root)
    hip, _ = load_hip_table(depth)
    fails = []
    for case in G.cases(depth):
        fails += G.run_case(hip, case, store, record=False)
    assert not fails, "\n".join(fails[:30])
