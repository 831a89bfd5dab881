"""CPU: the host-side float logic of stages.WeightAnalysis (LookaheadTLD::weightsAnalyse's guess, clamps, denominator reduction and
acceptance test) with the two device calls replaced by the oracle's weightCostLuma - the decision must equal the oracle's own
restatement of weightsAnalyse, which tests/test_oracle_classes_vs_reference.py pins against the real class.  (The device calls
themselves are covered by tests/test_gpu_lookahead_weights.py.)"""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


class HostTensor:
    """The two tensor methods WeightAnalysis uses, over a numpy array."""
    def __init__(self, a): self.a = a
    def cpu(self): return self
    def numpy(self): return self.a
    def __getitem__(self, k): return HostTensor(self.a[k])


@pytest.mark.parametrize("depth,gain,lift", [(8, 0.75, 6), (8, 1.0, 0), (8, 1.3, -20), (8, 0.5, 40), (8, 1.0, 9), (10, 0.8, 12), (10, 1.15, -6), (8, 0.25, 150),
                                             (8, 0.97, 1), (8, 2.4, -100), (10, 0.1, 90)])
def test_weight_analysis_host_logic_equals_oracle(depth, gain, lift, monkeypatch):
    import oracle_api as O
    width, height = 256, 144
    y0 = F.synth_clip(width, height, 1, depth=depth, seed=107)[0][0]
    pmax = (1 << depth) - 1
    noise = np.random.default_rng(107).integers(-1, 2, size=y0.shape) * (1 << (depth - 8))
    y1 = np.clip(np.rint(y0.astype(np.float64) * gain + lift * (1 << (depth - 8))) + noise, 0, pmax).astype(y0.dtype)
    cur, stride, org, w64, h64 = F.pad_plane(y1)
    ref = F.pad_plane(y0)[0]
    wcu, hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
    lw, lh = wcu * 8, hcu * 8
    rstride = (lw + 2 * F.MARGIN_X + 31) & ~31
    rows = lh + 2 * F.MARGIN_Y
    lorg = rstride * F.MARGIN_Y + F.MARGIN_X
    cp = O.lowres_init(depth, cur, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    rp = O.lowres_init(depth, ref, stride, org, rstride, lorg, rows, lw, lh, F.MARGIN_X, F.MARGIN_Y)
    icost, _, _ = O.lowres_intra(depth, cp[0], rstride, lorg, wcu, hcu, 5 if depth == 8 else 80)

    def stats(plane):
        a = plane.reshape(rows, rstride)[F.MARGIN_Y:F.MARGIN_Y + lh, F.MARGIN_X:F.MARGIN_X + lw].astype(np.int64)
        sm = int(a.sum())
        return int((a * a).sum()) - sm * sm // a.size, sm
    (ssd_c, sum_c), (ssd_r, sum_r) = stats(cp[0]), stats(rp[0])
    want = O.weights_analyse(depth, cp[0], rp[0], rstride, lorg, lw, lh, icost, (ssd_c, ssd_r), (sum_c, sum_r))

    applied = {}

    def fake_cost(d, fenc, refp, st, og, w, h, ic, cands, cost, stream=None):
        cost.a[:len(cands)] = np.array([O.lowres_weight_cost(d, fenc, refp, st, og, w, h, ic, c) for c in cands], np.uint32).view(np.int32)

    def fake_apply(d, src, dst, st, nrows, weight, stream=None):
        applied["weight"] = tuple(int(v) for v in weight)

    monkeypatch.setattr(S.hipabi, "lowres_weight_cost", fake_cost)
    monkeypatch.setattr(S.hipabi, "lowres_weight_apply", fake_apply)
    la = SimpleNamespace(depth=depth, planes=cp, stride=rstride, org=lorg, width=lw, lines=lh, intra_cost=icost, my=F.MARGIN_Y)
    lr = SimpleNamespace(planes=rp)
    wa = S.WeightAnalysis.__new__(S.WeightAnalysis)
    wa.la, wa.depth, wa.cost, wa.weighted = la, depth, HostTensor(np.zeros(4, np.int32)), [None] * 4
    got = wa.analyse(la, lr, (ssd_c, ssd_r), (sum_c, sum_r))
    assert got == want, f"host logic {got}, oracle {want}"
    assert (got[0] is None) == ("weight" not in applied) and (got[0] is None or applied["weight"] == got[0])
