"""Golden vectors: reference outputs recorded from the real reference build (tools/gen_golden.py,
committed under tests/golden/) replayed on the oracle table (CPU) and on the HIP table (GPU)."""
import os
import sys

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


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_weight_analyse_reproduces_the_weights_the_reference_chose(depth):
    """tests/golden/weight_analyse_d*.npz: slices of real encodes of fading clips - what x265's own weightAnalyse (encoder/weightPrediction.cpp:222) was handed and what
    it answered (tools/gen_weight_golden.py).  The oracle's restatement must choose the same weights; needs neither the reference nor a GPU."""
    import weight_fixture as WF
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api as O
    cs = WF.cases(depth)
    assert len(cs) >= 5 and any(c["nlists"] == 2 for c in cs) and any(c["expected"][0, 0, 0] for c in cs) and any(not c["expected"][0, 0, 0] for c in cs)
    for i, c in enumerate(cs):
        got, den = O.weight_analyse(depth, c["cur"], c["refs"], c["pic"][0], c["pic"][1], c["intra"])
        assert np.array_equal(got[:c["nlists"]], c["expected"][:c["nlists"]]), (i, got.tolist(), c["expected"].tolist())
        assert [int(den[l, 0]) for l in range(c["nlists"])] == [int(c["expected"][l, 0, 2]) for l in range(c["nlists"])]


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_aq_frame_reproduces_what_the_reference_left_in_lowres(depth):
    """tests/golden/aq_frame_d*.npz: the source picture x265's own calcAdaptiveQuantFrame was handed inside a real encode and the arrays it filled (AQ modes 1 - 3,
    qg 16 / 8, weightp on / off; tools/gen_weight_golden.py).  Doubles compared bit for bit."""
    import weight_fixture as WF
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api as O
    cs = WF.aq_cases(depth)
    assert len(cs) == 5 and {c["mode"] for c in cs} == {1, 2, 3} and {c["qg"] for c in cs} == {16, 8}
    for i, c in enumerate(cs):
        energy, qp, inv, sm, ssd = O.aq_frame(depth, c["y"], c["stride"], c["org"], c["width"], c["height"], cb=c["cb"], cr=c["cr"], stride_c=c["stride_c"], org_c=c["org_c"],
                                              qg_size=c["qg"], aq_mode=c["mode"], aq_strength=c["strength"], weightp=c["weightp"])
        assert np.array_equal(qp, c["qp_aq_offset"]) and np.array_equal(qp, c["qp_cutree_offset"]), i
        assert np.array_equal(inv, c["inv_qscale"]) and np.array_equal(sm, c["wp_sum"]) and np.array_equal(ssd, c["wp_ssd"]), i
