"""GPU parity: batched pixel-compare kernels (x265hip_pixelcmp_batch) vs the oracle table slots
(sad / satd / sa8d / sse_pp / psy_cost_pp), every PU / CU size, 8- and 10-bit, random + extremes."""
import importlib

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
spec = H.spec


def _cases(depth, kind):
    if kind in ("sad", "satd"):
        return [(f"pu[{i}].{kind}", *spec.pu_dims(i)) for i in range(25)]
    if kind == "sa8d":
        out = [(f"cu[{i}].sa8d", n, n) for i, n in enumerate(spec.LUMA_CU)]
        out += [("chroma[2].cu[2].sa8d", 8, 16), ("chroma[2].cu[3].sa8d", 16, 32), ("chroma[2].cu[4].sa8d", 32, 64)]
        return out
    if kind == "sse_pp":
        return [(f"cu[{i}].sse_pp", n, n) for i, n in enumerate(spec.LUMA_CU)] + [("chroma[2].cu[2].sse_pp", 8, 16)]
    return [(f"cu[{i}].psy_cost_pp", n, n) for i, n in enumerate(spec.LUMA_CU)]


KINDS = {"sad": A.CMP_SAD, "satd": A.CMP_SATD, "sa8d": A.CMP_SA8D, "sse_pp": A.CMP_SSE_PP, "psy_cost_pp": A.CMP_PSY_COST}


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("kind", list(KINDS))
def test_pixelcmp_batch_matches_oracle(depth, kind, repo_root):
    import torch
    dev = torch.device("cuda:0")
    orc = H.load_oracle(depth, repo_root)
    rng = np.random.default_rng([7, depth, KINDS[kind]])
    dt = H.pix_dtype(depth)
    m = H.pixel_max(depth)
    for path, w, h in _cases(depth, kind):
        fn = orc.fn(path)
        assert fn is not None, path
        sa, sb = 64, int(rng.integers(w + 16, w + 90))
        njobs = 37
        rows = h + 24
        a = rng.integers(0, m + 1, size=(njobs * rows * sa + 256)).astype(dt)
        b = rng.integers(0, m + 1, size=(rows * sb * 4 + 4096)).astype(dt)
        # TestBench-style extremes in some jobs: all-min vs all-max
        a[:rows * sa] = 0
        a_off = np.arange(njobs, dtype=np.int64) * rows * sa
        b_off = rng.integers(0, rows * sb * 3, size=njobs).astype(np.int64)    # arbitrary (unaligned) positions
        b_ext = b.copy()
        ta = torch.from_numpy(a.view(np.int16) if depth > 8 else a).to(dev)
        tb = torch.from_numpy(b_ext.view(np.int16) if depth > 8 else b_ext).to(dev)
        out = torch.zeros(njobs, dtype=torch.int64, device=dev)
        A.pixelcmp_batch(KINDS[kind], depth, w, h, ta, sa, tb, sb, njobs, out,
                         a_off=torch.from_numpy(a_off).to(dev), b_off=torch.from_numpy(b_off).to(dev))
        got = out.cpu().numpy()
        exp = np.array([fn(H.ptr(a, int(a_off[j])), sa, H.ptr(b, int(b_off[j])), sb) for j in range(njobs)], dtype=np.int64)
        assert np.array_equal(got, exp), f"{path} depth {depth}: {got[:6]} vs {exp[:6]}"


def test_pixelcmp_regular_step_and_empty(repo_root):
    import torch
    dev = torch.device("cuda:0")
    orc = H.load_oracle(8, repo_root)
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, size=64 * 64 * 5).astype(np.uint8)
    b = np.full(64 * 64 * 5, 255, dtype=np.uint8)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.zeros(5, dtype=torch.int64, device=dev)
    A.pixelcmp_batch(A.CMP_SAD, 8, 64, 64, ta, 64, tb, 64, 5, out, a_step=4096, b_step=4096)
    fn = orc.fn("pu[4].sad")
    exp = [fn(H.ptr(a, 4096 * j), 64, H.ptr(b, 4096 * j), 64) for j in range(5)]
    assert out.cpu().tolist() == exp
    A.pixelcmp_batch(A.CMP_SAD, 8, 64, 64, ta, 64, tb, 64, 0, out)      # empty batch is a no-op
    with pytest.raises(A.X265HipError):
        A.pixelcmp_batch(A.CMP_SAD, 8, 5, 8, ta, 64, tb, 64, 1, out)     # invalid block size fails loudly
