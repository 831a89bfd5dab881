"""Randomised soak of the kernels rewritten in round 3 (the exhaustive search with block-major records + the packed sub-pel arithmetic, and) the two whose results hang on tie-breaking: the serial pass of the SAO decision
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


def me_case(rng, torch):
    """Exhaustive search with block-major records (X265HIP_SURF_PACKED_B) and the sub-pel stage with packed arithmetic on its minima - random
    sizes, ranges, depths (the packed record formats are 8-bit), subme levels, both sub-pel flavours - against the oracle."""
    import importlib
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    import oracle_api as O
    dev = torch.device("cuda:0")
    depth = int(rng.choice([8, 8, 8, 10, 12]))
    w, h = int(rng.integers(1, 5)) * 64, int(rng.integers(1, 4)) * 64
    r = int(rng.integers(1, 41))
    subme = int(rng.integers(0, 8))
    planes = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 1 << 30))
    clip = F.synth_clip(w, h, 2, depth=depth, seed=seed)
    y0 = clip[0][0].astype(np.float32)
    sh = np.roll(y0, (int(rng.integers(-2, 3)), int(rng.integers(-3, 4))), axis=(0, 1))
    mix = float(rng.uniform(0.2, 0.8))
    y1 = np.clip(np.rint(mix * y0 + (1 - mix) * sh + rng.normal(0, 1.0, size=y0.shape)), 0, (1 << depth) - 1).astype(clip[0][0].dtype)
    cur, ref = P.DevicePicture(y1, dev), P.DevicePicture(clip[0][0], dev)
    fmt = "b" if depth == 8 else False
    ms = P.MotionSearch(cur.w64, cur.h64, r, depth, dev, packed=fmt)
    ms.run(cur, ref)
    sp = P.SubpelRefine(ms, subme, dev, phase_planes=planes)
    sp.run(cur, ref)
    torch.cuda.synchronize()
    desc = f"me {w}x{h} depth {depth} range {r} subme {subme} planes {planes} seed {seed}"
    surf, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, r, 0, ms.nctu, ms.cost_host, ms.cost_host)
    gb = ms.best.cpu().numpy().view(np.uint64)
    if not np.array_equal(gb, best):
        return f"{desc}: best differs ({np.count_nonzero(gb != best)})"
    e = surf.reshape(ms.nctu * ms.nc, ms.ng, 85, 4).transpose(0, 1, 3, 2).reshape(ms.nctu * ms.nc, ms.ng * 4, 85)[:, :ms.nc, :]
    for level in range(4):
        b, n = P.LEVEL_BASE[level], P.LEVEL_PUS[level]
        g = ms.level_view(level)[0].cpu().numpy()
        if not np.array_equal(g, e[:, :, b:b + n].reshape(-1, n)):
            return f"{desc}: surface level {level} differs"
    exp = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, r, 0, ms.nctu, best, sp.cost_q_host, sp.qoff, subme)
    got = sp.out.cpu().numpy().reshape(-1, 2)
    if not np.array_equal(got, exp):
        return f"{desc}: sub-pel output differs for {int((got != exp).any(axis=1).sum())} PUs"
    return None


def pipe_case(rng, torch):
    """The whole frame pipeline as bench.py runs it - lookahead, exhaustive search, sub-pel stage, luma + chroma TU stages with sign hiding,
    deblocking, SAO statistics, the reference's SAO decision, application, borders - on a random small picture, with the round-3 launch
    schedule (parallel_planes) or on one stream, every stage output against the oracle chain (bench.py's own bit_exact comparison)."""
    import importlib
    sys.path.insert(0, ROOT)
    import bench as B
    from test_gpu_pipeline import _sao_rdo_inputs
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    import oracle_api as O
    dev = torch.device("cuda:0")
    depth = int(rng.choice([8, 8, 10]))
    w, h = int(rng.integers(2, 7)) * 64 - int(rng.choice([0, 0, 24])), int(rng.integers(1, 5)) * 64 - int(rng.choice([0, 0, 16]))
    R, subme = int(rng.integers(4, 21)), int(rng.integers(0, 8))
    qp = int(rng.integers(16, 42)) + 12 * (depth == 10)
    par = bool(rng.integers(0, 2))
    fmt = [True, "t", "b"][int(rng.integers(0, 3))] if depth == 8 else False
    seed = int(rng.integers(0, 1 << 30))
    clip = F.synth_clip(w, h, 2, depth=depth, seed=seed)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    srdo = _sao_rdo_inputs(depth, qp)
    pipe = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=R, subme=subme, level=2, qp=qp, want_surf=True, packed=fmt, lookahead=(w, h),
                           deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True, sao_rdo=srdo, subpel_planes=bool(rng.integers(0, 2)),
                           parallel_planes=par)
    dev_out = B.device_outputs(pipe, pics[1], pics[0])
    if par:                                            # a second frame through the same pipeline object: the double-buffered minima, the ping-pong
        new = pipe.swap_output([p.clone() for p in pics[0].planes()])
        dev_out = B.device_outputs(pipe, pics[1], pics[0])
    _, cpu_out = B.oracle_chain(F, clip, R, subme, 2, qp, depth, pipe.ms.nctu, B.effective_cpus(), O.host_has_avx2(), sao_rdo=srdo)
    res = B.compare_outputs(dev_out, cpu_out)
    if not res["ok"]:
        bad = {k: v for k, v in res["stages"].items() if v != "equal"}
        return f"pipeline {w}x{h} depth {depth} range {R} subme {subme} qp {qp} parallel {par} format {fmt} seed {seed}: {bad}"
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
    n = {"sao": 0, "lookahead": 0, "me": 0, "pipe": 0}
    fails = []
    t0 = time.time()
    while time.time() - t0 < args.seconds and len(fails) < 5:
        u = rng.random()
        if u < 0.55:
            r = sao_case(rng, A, HT, O, tabs, dev, torch); n["sao"] += 1
        elif u < 0.75:
            r = la_case(rng, torch); n["lookahead"] += 1
        elif u < 0.9:
            r = me_case(rng, torch); n["me"] += 1
        else:
            r = pipe_case(rng, torch); n["pipe"] += 1
        if r:
            fails.append(r)
            print("MISMATCH", r, flush=True)
    print(f"r3_soak seed {args.seed}: {n['sao']} SAO decisions + {n['lookahead']} lowres intra estimates + {n['me']} search / sub-pel pictures + {n['pipe']} whole-pipeline frames in {time.time() - t0:.0f} s, {len(fails)} mismatches")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
