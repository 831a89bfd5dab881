#!/usr/bin/env python3
"""[test infrastructure: uses oracle/ only as the checker; not collected by pytest - run it by hand on a GPU box]

Randomised soak of the device stages against the oracle: many seeds / sizes / parameter draws per stage for a time budget.
Usage on a GPU box:  python tests/soak.py [seconds [master seed]]   -> one line per stage with the number of cases, non-zero exit on a mismatch."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import oracle_api as O
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
import test_gpu_search as TS
import test_gpu_deblock as TD
import test_gpu_intra_recon as TI
import test_gpu_aq as TA
import test_gpu_lookahead_weights as TW
from test_oracle_classes_vs_reference import sao_case

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
master = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260927)


def soak_search(rng):
    depth = int(rng.choice([8, 10, 12]))
    method = str(rng.choice(["dia", "hex", "umh", "star", "full"]))
    w, h = int(rng.choice([128, 192, 320])), int(rng.choice([128, 192]))
    TS._check(depth, method, w, h, int(rng.integers(1, 1 << 30)), 96 if method != "full" else 24, [int(rng.integers(0, 8))],
              [int(rng.choice([4, 8])) if method == "full" else int(rng.choice([8, 16, 32, 57]))])


def soak_lowres(rng):
    depth = int(rng.choice([8, 10]))
    w, h = int(rng.choice([96, 208, 256, 400])), int(rng.choice([64, 144, 208]))
    seed = int(rng.integers(1, 1 << 30))
    clip = F.synth_clip(w, h, 3, depth=depth, seed=seed)
    ys = [clip[i][0] for i in range(3)]
    ys[1] = np.roll(ys[1], (int(rng.integers(-3, 4)), int(rng.integers(-5, 6))), axis=(0, 1))
    pics = [P.DevicePicture(y, dev) for y in ys]
    las = [S.Lookahead(w, h, depth, dev, intra_penalty=5 if depth == 8 else 80) for _ in range(3)]
    for la, pic in zip(las, pics):
        la.run(pic)
    bidir = bool(rng.integers(0, 2))
    st = S.LookaheadCost(las[1], dev, bidir=bidir)
    bias = int(rng.integers(0, 40))
    st.run(las[1], las[0], las[2] if bidir else None, bframe_bias=bias)
    torch.cuda.synchronize()
    dt = ys[0].dtype
    planes = [[p.cpu().numpy().view(dt) for p in la.planes] for la in las]
    cq = st.cost_q.cpu().numpy().view(np.uint16)
    res = O.lowres_cost(depth, planes[1][0], planes[0], las[1].stride, las[1].org, las[1].wcu, las[1].hcu, cq, st.qoff, las[1].intra_cost.cpu().numpy(),
                        ref1_planes=planes[2] if bidir else None, bframe_bias=bias)
    if bidir:
        assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), res[0][0]) and np.array_equal(st.mvs1.cpu().numpy().reshape(-1, 2), res[0][1])
        assert np.array_equal(st.frame.cpu().numpy(), res[4])
    else:
        assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), res[0]) and np.array_equal(st.frame.cpu().numpy()[:3], res[4])
    assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), res[2]) and np.array_equal(st.row_satds.cpu().numpy(), res[3])


def soak_sao(rng):
    depth = int(rng.choice([8, 10, 12]))
    w, h = int(rng.integers(17, 300)) * 2, int(rng.integers(17, 200)) * 2
    y, rec, params = sao_case(depth, w, h, int(rng.integers(1, 1 << 30)))
    fenc, stride, org, _, _ = F.pad_plane(y)
    recp = F.pad_plane(rec)[0]
    nctu = params.shape[0]
    d_f = torch.from_numpy(fenc.view(np.uint8).reshape(-1)).to(dev)
    d_r = torch.from_numpy(recp.view(np.uint8).reshape(-1)).to(dev)
    d_c = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
    d_o = torch.zeros(nctu * 160, dtype=torch.int32, device=dev)
    A.sao_stats(depth, d_f, stride, org, d_r, stride, org, w, h, d_c, d_o)
    d_out = d_r.clone()
    A.sao_apply(depth, d_r, stride, org, d_out, stride, org, w, h, torch.from_numpy(params.reshape(-1)).to(dev))
    torch.cuda.synchronize()
    cnt, off = O.sao_stats(depth, fenc, recp, stride, org, w, h)
    assert np.array_equal(d_c.cpu().numpy().reshape(cnt.shape), cnt) and np.array_equal(d_o.cpu().numpy().reshape(off.shape), off)
    out = O.sao_apply(depth, recp, stride, org, w, h, params)
    assert np.array_equal(d_out.cpu().numpy().view(y.dtype).reshape(out.shape), out)


def soak_pipeline(rng):
    depth = int(rng.choice([8, 10]))
    level, subme = int(rng.integers(0, 3)), int(rng.integers(0, 8))
    qp = int(rng.integers(10, 46)) + 12 * (depth == 10)
    R = int(rng.choice([6, 12, 20]))
    clip = F.synth_clip(192, 128, 3, depth=depth, seed=int(rng.integers(1, 1 << 30)))
    pics = [P.DevicePicture(y, dev) for (y, _, _) in clip]
    fp = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=R, subme=subme, level=level, qp=qp, want_surf=False, deblock=True, sao=True)
    cost = F.mv_cost_table(R)
    cq, qoff = F.qpel_cost_table(R)
    cur, ref = pics[1], pics[0]
    rec = fp.run(cur, ref)
    torch.cuda.synchronize()
    _, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, cur.stride, cur.org, cur.w64, cur.h64, R, 0, fp.ms.nctu, cost, cost, want_surf=False)
    mv = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref.host, cur.stride, cur.org, cur.w64, cur.h64, R, 0, fp.ms.nctu, best, cq, qoff, subme)
    erec, elev, ens, _ = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, cur.stride, cur.org, cur.w64, cur.h64, level, mv, qp)
    bv, bh = O.deblock_bs_inter(depth, cur.w64, cur.h64, level, mv, ens)
    erec = O.deblock_luma(depth, erec.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64, bv, bh, max(qp - 6 * (depth - 8), 0)).reshape(erec.shape)
    ecnt, eoff = O.sao_stats(depth, cur.host.reshape(-1), erec.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64)
    assert np.array_equal(fp.sp.out.cpu().numpy().reshape(-1, 2), mv) and np.array_equal(fp.rc.levels.cpu().numpy(), elev)
    inner = erec[F.MARGIN_Y:F.MARGIN_Y + cur.h64, F.MARGIN_X:F.MARGIN_X + cur.w64]
    erec = np.pad(inner, ((F.MARGIN_Y, F.MARGIN_Y), (F.MARGIN_X, F.MARGIN_X)), mode="edge")
    assert np.array_equal(rec.cpu().numpy().view(cur.host.dtype).reshape(cur.host.shape), erec)
    assert np.array_equal(fp.sao.count.cpu().numpy().reshape(ecnt.shape), ecnt) and np.array_equal(fp.sao.offset_org.cpu().numpy().reshape(eoff.shape), eoff)


def soak_deblock(rng):
    depth = int(rng.choice([8, 10, 12]))
    TD.test_deblock_with_intra_blocks_and_chroma(depth, int(rng.integers(0, 3)), int(rng.integers(26, 48)), (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))))


def soak_intra_tu(rng):
    depth = int(rng.choice([8, 10, 12]))
    TI.test_intra_recon_matches_oracle(depth, int(rng.choice([4, 8, 16, 32])), int(rng.integers(0, 52)) + 6 * (depth - 8), int(rng.integers(0, 2)),
                                       bool(rng.integers(0, 2)))


def soak_sea(rng):
    depth = int(rng.choice([8, 10, 12]))
    TS._check_sea(depth, int(rng.choice([192, 256, 320])), int(rng.choice([192, 256])), int(rng.integers(1, 1 << 30)), 64,
                  [int(rng.integers(0, 8))], [int(rng.choice([4, 9, 16, 33, 40]))])


def soak_bs(rng):
    TD.test_boundary_strengths_multi_reference_and_b_pictures(int(rng.integers(0, 4)), int(rng.integers(0, 2)), str(rng.choice(["both", "noref1", "none"])), seed=int(rng.integers(1, 1 << 30)))


def soak_aq(rng):
    depth = int(rng.choice([8, 10]))
    qg = int(rng.choice([16, 8]))
    w, h = int(rng.choice([128, 250, 416, 640])), int(rng.choice([96, 138, 240]))
    chroma = bool(rng.integers(0, 2)) and w % 16 == 0 and h % 16 == 0
    TA.test_aq_pass_matches_oracle(depth, w, h, qg, int(rng.integers(1, 4)), float(rng.choice([0.5, 1.0, 1.7])), chroma, seed=int(rng.integers(1, 1 << 30)))


def soak_weights(rng):
    depth = int(rng.choice([8, 10]))
    gain, lift = float(rng.choice([0.3, 0.6, 0.8, 0.95, 1.1, 1.4])), int(rng.integers(-40, 60))
    TW.test_weight_analysis_matches_oracle(depth, int(rng.choice([256, 384, 512])), int(rng.choice([128, 256])), gain, lift, seed=int(rng.integers(1, 1 << 30)), check_expectation=False)


stages = [("SEA search", soak_sea), ("boundary strengths (multi-ref / B)", soak_bs), ("adaptive quantisation", soak_aq), ("weight analysis", soak_weights),
          ("deblocking (intra / chroma)", soak_deblock), ("intra TU candidates", soak_intra_tu), ("search drivers", soak_search), ("lookahead cost (P/B)", soak_lowres), ("sao passes", soak_sao), ("frame pipeline", soak_pipeline)]
counts = {n: 0 for n, _ in stages}
t0 = time.time()
fail = 0
while time.time() - t0 < budget and not fail:
    for name, fn in stages:
        seed = int(master.integers(1, 1 << 31))
        try:
            fn(np.random.default_rng(seed))
            counts[name] += 1
        except AssertionError as e:
            print(f"MISMATCH in {name} (case seed {seed}): {str(e)[:400]}", flush=True)
            fail = 1
            break
for n, c in counts.items():
    print(f"{n}: {c} randomised cases matched the oracle")
print(f"soak {'FAILED' if fail else 'ok'} after {time.time() - t0:.0f} s")
sys.exit(fail)
