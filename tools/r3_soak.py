"""Randomised soak of the two kernels rewritten in round 3 whose results hang on tie-breaking: the serial pass of the SAO decision
(x265hip_sao_rdo, second organisation) and the mode-parallel lowres intra estimate - device against oracle, fresh random cases for a
time budget.  SAO: random / run-copied / sparse / extreme statistics (copied runs make the merge candidates tie with the new parameters),
1..40 x 1..70 CTUs (one and two wavefronts of rows; the fall-back to the first organisation beyond 76 rows), 8/10/12-bit, luma-only and
4:2:0, either SAO flag off, per-CTU lambdas, all slice types.  Lookahead: random sizes, both depths, textured / flat / extreme content.

  python tools/r3_soak.py --seconds 120 [--seed 1]        (on the GPU box; measurement aid, not part of the test suite)"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sao_case(rng, A, HT, O, tabs, dev, torch):
    depth = int(rng.choice([8, 8, 10, 12]))
    cw, ch = int(rng.integers(1, 41)), int(rng.choice([1, 2, 3, 5, 17, 34, 37, 40, 64, 65, 68, 76, 80]))
    planes = int(rng.choice([1, 3, 3]))
    nctu = cw * ch
    mode = int(rng.integers(0, 4))
    maxv = (1 << depth) - 1
    cnt, org = [], []
    for pl in range(planes):
        area = 4096 if pl == 0 else 1024
        if mode == 2:                                         # sparse
            c = (rng.random((nctu, 5, 32)) < 0.05) * rng.integers(1, area // 8, size=(nctu, 5, 32))
        else:
            c = rng.integers(0, area // (4 if mode else 16), size=(nctu, 5, 32))
        scale = maxv if mode == 3 else max(2, maxv // 64)     # extreme: sums as large as the samples allow
        o = (rng.standard_normal((nctu, 5, 32)) * c * rng.uniform(0.1, 1.0) * scale / 8).astype(np.int64)
        o = np.clip(o, -c * maxv, c * maxv)
        if mode == 1:                                         # runs of CTUs with the same statistics: merge candidates tie with new parameters
            keep = rng.random(nctu) < 0.35
            keep[0] = True
            src = np.maximum.accumulate(np.where(keep, np.arange(nctu), 0))
            up = rng.random(nctu) < 0.3
            src = np.where(up & (np.arange(nctu) >= cw), src[np.maximum(np.arange(nctu) - cw, 0)], src)
            c, o = c[src], o[src]
        cnt.append(np.ascontiguousarray(c, dtype=np.int32))
        org.append(np.ascontiguousarray(o, dtype=np.int32))
    qp = int(rng.integers(10, 46))
    per_ctu = rng.random() < 0.4
    ctu_qp = np.clip(qp + rng.integers(-4, 5, size=nctu), 0, 51) if per_ctu else np.full(nctu, qp)
    lam = np.array([HT.sao_lambdas(tabs, int(q), csp400=planes == 1) for q in ctu_qp], dtype=np.int64)
    st = int(rng.integers(0, 3))
    ctx_m, ctx_t = HT.sao_contexts(st, qp)
    flag = (1, 1) if planes == 1 else [(1, 1), (1, 1), (1, 0), (0, 1)][int(rng.integers(0, 4))]
    flag = (flag[0], flag[1] if planes == 3 else 0)
    if flag == (0, 0):
        flag = (1, 0)
    dc = [torch.from_numpy(c.reshape(-1)).to(dev) for c in cnt]
    do = [torch.from_numpy(o.reshape(-1)).to(dev) for o in org]
    par = [torch.full((nctu * 7,), 0x5a5a5a5a, dtype=torch.int32, device=dev) for _ in range(planes)]
    scratch = torch.zeros(A.sao_rdo_scratch_bytes(cw, ch), dtype=torch.uint8, device=dev)
    nos = torch.full((2,), -1, dtype=torch.int32, device=dev)
    A.sao_rdo(depth, dc, do, cw, ch, lam[0], ctx_m, ctx_t, tabs["entropy_bits"], par, scratch,
              lambda_ctu=torch.from_numpy(lam).to(dev) if per_ctu else None, sao_flag=flag, num_no_sao=nos)
    torch.cuda.synchronize()
    ep, en = O.sao_rdo(depth, cnt, org, cw, ch, lam, ctx_m, ctx_t, tabs["entropy_bits"], sao_flag=flag)
    desc = f"sao depth {depth} {cw}x{ch} planes {planes} mode {mode} qp {qp} per_ctu {per_ctu} slice {st} flag {flag}"
    for pl in range(planes):
        got = par[pl].cpu().numpy().reshape(nctu, 7)
        if not np.array_equal(got, ep[pl]):
            bad = np.argwhere((got != ep[pl]).any(axis=1))[:4].reshape(-1).tolist()
            return f"{desc}: plane {pl} CTUs {bad}: device {got[bad].tolist()} oracle {ep[pl][bad].tolist()}"
    gn = nos.cpu().numpy()
    if int(gn[0]) != int(en[0]) or (planes == 3 and flag[1] and int(gn[1]) != int(en[1])):
        return f"{desc}: numNoSao device {gn.tolist()} oracle {en.tolist()}"
    return None


def la_case(rng, torch):
    import test_gpu_lookahead as T
    depth = int(rng.choice([8, 10]))
    w, h = int(rng.integers(4, 160)) * 16, int(rng.integers(4, 90)) * 8
    extreme = [None, None, "noise", "max"][int(rng.integers(0, 4))]
    try:
        T._run(w, h, depth, seed=int(rng.integers(0, 1 << 30)), extreme=extreme, penalty=int(rng.choice([0, 5, 40])))
    except AssertionError as e:
        return f"lookahead {w}x{h} depth {depth} {extreme}: {str(e)[:300]}"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")
    import oracle_api as O
    tabs = HT.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    n = {"sao": 0, "lookahead": 0}
    fails = []
    t0 = time.time()
    while time.time() - t0 < args.seconds and len(fails) < 5:
        if rng.random() < 0.75:
            r = sao_case(rng, A, HT, O, tabs, dev, torch); n["sao"] += 1
        else:
            r = la_case(rng, torch); n["lookahead"] += 1
        if r:
            fails.append(r)
            print("MISMATCH", r, flush=True)
    print(f"r3_soak seed {args.seed}: {n['sao']} SAO decisions + {n['lookahead']} lowres intra estimates in {time.time() - t0:.0f} s, {len(fails)} mismatches")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
