"""GPU parity: one cuTree propagation step (x265hip_cutree_propagate) vs the oracle's restatement of Lookahead::estimateCUPropagate +
primitives.propagateCost (oracle/x265_oracle_pipeline3.c), which tests/test_oracle_classes_vs_reference.py pins against the real
Lookahead on motion fields its own frame cost estimate produced."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


def make_case(wcu, hcu, bframe, referenced, seed):
    """A random but plausible step: coherent mvs (some pointing out of the picture), costs with the lists-used bits, a few blocks
    with intra cost 0 (0 / 0 in the reference's double arithmetic: nothing is passed on) and intra <= inter (nothing either),
    references partly about to saturate."""
    rng = np.random.default_rng([61, wcu, hcu, int(bframe), int(referenced), seed])
    n = wcu * hcu
    intra = rng.integers(200, 9000, size=n).astype(np.int32)
    intra[rng.random(n) < 0.02] = 0
    inter = (intra * rng.random(n) * 1.2).astype(np.int64).clip(0, 16383)
    lists = rng.choice([1, 2, 3], size=n, p=[0.5, 0.2, 0.3]) if bframe else rng.integers(0, 2, size=n)
    lcost = (inter | (lists << 14)).astype(np.uint16)
    def mvfield():
        base = rng.integers(-40, 41, size=(hcu // 4 + 1, wcu // 4 + 1, 2))
        m = np.kron(base, np.ones((4, 4, 1), np.int64))[:hcu, :wcu].reshape(n, 2) + rng.integers(-3, 4, size=(n, 2))
        m[rng.random(n) < 0.2] = 0
        m[rng.random(n) < 0.03] *= 12                                   # far out of the picture
        return m.astype(np.int32)
    mv0, mv1 = mvfield(), mvfield()
    invq = rng.integers(64, 1024, size=n).astype(np.int32)
    prop = rng.integers(0, 40000, size=n).astype(np.uint16) if referenced else None
    r0 = rng.integers(0, 65536, size=n).astype(np.uint16); r0[::3] = 65520
    r1 = rng.integers(0, 30000, size=n).astype(np.uint16)
    return intra, lcost, mv0, mv1, invq, prop, r0, r1


@pytest.mark.parametrize("wcu,hcu,bframe,referenced,fps,bipred", [(240, 135, True, True, 1.0, 32), (240, 135, False, True, 0.8, 32), (120, 68, True, False, 1.25, 40),
                                                                  (17, 9, True, True, 0.02, 17), (3, 2, False, False, 1.0, 32), (64, 36, True, True, 4.0, 32)])
def test_cutree_step_matches_oracle(wcu, hcu, bframe, referenced, fps, bipred):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    intra, lcost, mv0, mv1, invq, prop, r0, r1 = make_case(wcu, hcu, bframe, referenced, 1)
    e0, e1 = O.cutree_propagate(8, wcu, hcu, prop, intra, lcost, invq, mv0, mv1 if bframe else None, fps, bipred, r0, r1 if bframe else None)
    u16 = lambda a: None if a is None else torch.from_numpy(a.view(np.int16).copy()).to(dev)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1).copy()).to(dev)
    d0, d1 = u16(r0), u16(r1) if bframe else None
    A.cutree_propagate(wcu, hcu, u16(prop), i32(intra), u16(lcost), i32(invq), i32(mv0), i32(mv1) if bframe else None, fps, bipred, d0, d1)
    torch.cuda.synchronize()
    g0 = d0.cpu().numpy().view(np.uint16)
    assert np.array_equal(g0, e0), f"list-0 reference: {np.count_nonzero(g0 != e0)} of {g0.size} blocks differ"
    if bframe:
        g1 = d1.cpu().numpy().view(np.uint16)
        assert np.array_equal(g1, e1), f"list-1 reference: {np.count_nonzero(g1 != e1)} of {g1.size} blocks differ"
    if wcu * hcu > 100:
        assert (e0 != r0).any() and (e0 == 65535).any()


def test_cutree_rejects_bad_arguments():
    import torch
    dev = torch.device("cuda:0")
    z32, z16 = torch.zeros(64, dtype=torch.int32, device=dev), torch.zeros(64, dtype=torch.int16, device=dev)
    with pytest.raises(A.X265HipError):
        A.cutree_propagate(4, 4, None, z32, z16, z32, z32, z32, 1.0, 32, z16, None)        # mvs1 without ref_cost1
    with pytest.raises(A.X265HipError):
        A.cutree_propagate(4, 4, None, z32, z16, z32, z32, None, 1.0, 99, z16, None)
