"""GPU parity of the closed-loop frame pipeline (ME -> sub-pel -> prediction/residual round trip -> border
extension -> next reference) against the same chain of oracle stages, frame by frame."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")


def _oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api
    return oracle_api


@pytest.mark.parametrize("depth,deblock", [(8, False), (10, False), (8, True), (10, True)])
def test_closed_loop_three_frames(depth, deblock):
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    R, subme, level, qp = 12, 2, 2, (30 if deblock else 24) + 12 * (depth == 10)
    clip = F.synth_clip(192, 128, 4, depth=depth, seed=61)
    pics = [P.DevicePicture(y, dev) for (y, _, _) in clip]
    fp = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=R, subme=subme, level=level, qp=qp, want_surf=False, deblock=deblock, sao=deblock)
    ref_dev = P.DevicePicture(clip[0][0], dev)             # frame 0 is the first reference as-is
    ref_host = ref_dev.host.copy()
    cost = F.mv_cost_table(R)
    cq, qoff = F.qpel_cost_table(R)
    for k in (1, 2, 3):
        cur = pics[k]
        rec = fp.run(cur, ref_dev)
        torch.cuda.synchronize()
        # oracle chain on the host
        _, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref_host, cur.stride, cur.org, cur.w64, cur.h64, R,
                                  0, fp.ms.nctu, cost, cost, want_surf=False)
        mv = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref_host, cur.stride, cur.org, cur.w64, cur.h64, R,
                             0, fp.ms.nctu, best, cq, qoff, subme)
        erec, elev, ens, edist = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref_host, cur.stride, cur.org,
                                               cur.w64, cur.h64, level, mv, qp)
        if deblock:         # the in-loop filter runs on the reconstruction before it becomes a reference
            bv, bh = O.deblock_bs_inter(depth, cur.w64, cur.h64, level, mv, ens)
            erec = O.deblock_luma(depth, erec.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64, bv, bh, max(qp - 6 * (depth - 8), 0)).reshape(erec.shape)
        if deblock:         # SAO statistics of the filtered reconstruction against the source
            ecnt, eoff = O.sao_stats(depth, cur.host.reshape(-1), erec.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64)
            assert np.array_equal(fp.sao.count.cpu().numpy().reshape(ecnt.shape), ecnt), f"frame {k}: SAO counts differ"
            assert np.array_equal(fp.sao.offset_org.cpu().numpy().reshape(eoff.shape), eoff), f"frame {k}: SAO offset sums differ"
        inner = erec[F.MARGIN_Y:F.MARGIN_Y + cur.h64, F.MARGIN_X:F.MARGIN_X + cur.w64]
        erec = np.pad(inner, ((F.MARGIN_Y, F.MARGIN_Y), (F.MARGIN_X, F.MARGIN_X)), mode="edge")   # extendPicBorder
        grec = rec.cpu().numpy().view(cur.host.dtype).reshape(cur.host.shape)
        assert np.array_equal(fp.ms.best.cpu().numpy().view(np.uint64), best), f"frame {k}: integer mvs differ"
        assert np.array_equal(fp.sp.out.cpu().numpy().reshape(-1, 2), mv), f"frame {k}: sub-pel mvs differ"
        assert np.array_equal(fp.rc.levels.cpu().numpy(), elev), f"frame {k}: levels differ"
        assert np.array_equal(grec, erec), f"frame {k}: reconstruction (with borders) differs"
        # the reconstruction becomes the reference of the next frame on both sides
        ref_host = erec
        ref_dev = P.DevicePicture.__new__(P.DevicePicture)
        ref_dev.__dict__.update(cur.__dict__)
        ref_dev.t = rec.clone()
        ref_dev.host = erec


@pytest.mark.parametrize("depth", [8, 10])
def test_lookahead_costs_run_ahead_in_batches(depth):
    """FramePipeline(lookahead_cost_batch=2): the lookahead's P-frame cost estimates are launched two pictures at a time on a side
    stream over a ring of prepared pictures; every (picture, previous picture) pair must equal the oracle's estimate."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    W, Hh = 192, 128
    clip = F.synth_clip(W, Hh, 8, depth=depth, seed=63)
    pics = [P.DevicePicture(y, dev) for (y, _, _) in clip]
    fp = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=8, subme=1, level=2, qp=30, want_surf=False, lookahead=(W, Hh),
                         lookahead_cost_batch=2)
    dt = clip[0][0].dtype
    planes = []                                             # host copies of the prepared pictures, frame by frame

    def expect(k):
        la = fp.ring[0]
        cq = fp.lc[0].cost_q.cpu().numpy().view(np.uint16)
        cur_planes, ref_planes, icost = planes[k][0], planes[k - 1][0], planes[k][1]
        return O.lowres_cost(depth, cur_planes[0], ref_planes, la.stride, la.org, la.wcu, la.hcu, cq, fp.lc[0].qoff, icost)

    checked = 0
    ref = pics[0]
    for k in range(7):
        n_before = len(fp.pending)
        fp.run(pics[k], ref)
        torch.cuda.synchronize()
        planes.append(([p.cpu().numpy().view(dt).copy() for p in fp.la.planes], fp.la.intra_cost.cpu().numpy().copy()))
        if k and n_before + 1 == 2:                         # this frame completed a batch: frames k-1 and k were scored
            for i, kk in enumerate((k - 1, k)):
                mvs, mvc, lcost, rows, frame = expect(kk)
                st = fp.lc[i]
                assert np.array_equal(st.mvs.cpu().numpy().reshape(-1, 2), mvs), f"frame {kk}: lookahead mvs differ"
                assert np.array_equal(st.lowres_costs.cpu().numpy().view(np.uint16), lcost) and np.array_equal(st.frame.cpu().numpy()[:3], frame)
                checked += 1
    assert checked == 6 and not fp.pending
    fp.run(pics[7], ref)
    assert len(fp.pending) == 1
    fp.launch_lookahead_costs()                             # flush the incomplete batch
    torch.cuda.synchronize()
    planes.append(([p.cpu().numpy().view(dt).copy() for p in fp.la.planes], fp.la.intra_cost.cpu().numpy().copy()))
    mvs, mvc, lcost, rows, frame = expect(7)
    assert np.array_equal(fp.lc[0].mvs.cpu().numpy().reshape(-1, 2), mvs) and np.array_equal(fp.lc[0].frame.cpu().numpy()[:3], frame)


def _sao_rdo_inputs(depth, qp):
    """The host-side inputs of x265hip_sao_rdo for a P picture at quantiser QP `qp` (bench.py builds the same record)."""
    HT = importlib.import_module("x265-yuuki-asuna_amd.host_tables")
    tabs = HT.load()
    cu_qp = max(qp - 6 * (depth - 8), 0)
    cm, ct = HT.sao_contexts(HT.SLICE_P, cu_qp)
    return {"lambdas": HT.sao_lambdas(tabs, cu_qp), "ctx_merge": cm, "ctx_type": ct, "entropy_bits": tabs["entropy_bits"]}


@pytest.mark.parametrize("depth,fast,rdo", [(8, False, False), (10, False, False), (8, True, False), (10, True, False), (8, False, True), (8, True, True), (10, True, True)])
def test_closed_loop_with_chroma_and_sao_in_the_loop(depth, fast, rdo):
    """The default bench pipeline (luma + 4:2:0 chroma reconstruction, luma + chroma deblocking, SAO statistics -> on-device parameters
    -> SAO apply on Y / Cb / Cr, border extension) over three frames, each searched in and predicted from the previous frame's
    FILTERED reconstruction; every stage output of every frame against the oracle chain (bench.py's bit_exact code path).
    fast: bench.py's default launch structure - sub-pel candidates read from the reference's phase planes, the Cb / Cr chains and the
    lookahead on their own HIP streams.  rdo: the SAO parameters from x265hip_sao_rdo (the reference's rate-distortion decision, bench.py's
    default) instead of the distortion-only stand-in."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    O = _oracle()
    dev = torch.device("cuda:0")
    W, Hh, R, subme, level, qp = 256, 192, 12, 3, 2, 30 + 12 * (depth == 10)
    clip = F.synth_clip(W, Hh, 4, depth=depth, seed=67)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    srdo = _sao_rdo_inputs(depth, qp) if rdo else None
    pipe = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=R, subme=subme, level=level, qp=qp, want_surf=False, lookahead=(W, Hh),
                           deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True, subpel_planes=fast, parallel_planes=fast, sao_rdo=srdo)
    ref_dev = pics[0].like([p.clone() for p in pics[0].planes()])
    ref_host = None                                                    # frame 1 searches the source frame 0
    types = set()
    for k in (1, 2, 3):
        dev_out = B.device_outputs(pipe, pics[k], ref_dev)
        _, cpu_out = B.oracle_chain(F, clip, R, subme, level, qp, depth, pipe.ms.nctu, 4, False, ref_planes=ref_host, cur_index=k, sao_rdo=srdo)
        res = B.compare_outputs(dev_out, cpu_out)
        assert res["ok"], f"frame {k}: {res['stages']}"
        types |= set(np.asarray(cpu_out["sao_params"]).reshape(-1, 7)[:, 0].tolist()) | set(np.asarray(cpu_out["sao_params_c0"]).reshape(-1, 7)[:, 0].tolist())
        ref_host = (cpu_out["recon"], cpu_out["recon_c0"].reshape(-1), cpu_out["recon_c1"].reshape(-1))
        ref_dev = pics[k].like([p.clone() for p in pipe.final_planes()])
    assert any(t >= 0 for t in types), "SAO never switched on: the loop was not exercised"


@pytest.mark.parametrize("depth,split,packed", [(8, 2, "t"), (8, 3, True), (10, 2, False)])
def test_search_to_reconstruction_in_parts_leaves_the_same_picture(depth, split, packed):
    """FramePipeline(split=k): search -> refinement -> reconstruction in k parts of whole CTU rows, parts refined / reconstructed on a side
    stream while the next one is searched (stage objects' part() views).  Every stage output and the filtered planes equal the picture
    processed in one piece, over a closed loop of three frames."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(256, 320, 4, depth=depth, seed=64)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    kw = dict(rng=12, subme=3, level=2, qp=30 + 12 * (depth == 10), want_surf=True, packed=packed, lookahead=(256, 320), deblock=True, sao=True,
              chroma=True, sao_apply=True, sign_hide=True, subpel_planes=True, parallel_planes=True)
    whole = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, **kw)
    parts = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, split=split, **kw)
    refs = [pics[0].like([p.clone() for p in pics[0].planes()]) for _ in range(2)]
    for k in (1, 2, 3):
        for fp, ref in ((whole, refs[0]), (parts, refs[1])):
            fp.run(pics[k], ref)
            torch.cuda.synchronize()
        assert parts.parts is not None and len(parts.parts) == split
        for name in ("best", "surf"):
            assert torch.equal(getattr(whole.ms, name), getattr(parts.ms, name)), f"frame {k}: search {name} differs"
        assert torch.equal(whole.sp.out, parts.sp.out), f"frame {k}: refined vectors differ"
        for a, b in [(whole.rc, parts.rc)] + list(zip(whole.rc_c, parts.rc_c)):
            assert torch.equal(a.levels, b.levels) and torch.equal(a.num_sig, b.num_sig) and torch.equal(a.dist, b.dist), f"frame {k}: TU stage outputs differ"
        for i, (a, b) in enumerate(zip(whole.final_planes(), parts.final_planes())):
            assert torch.equal(a, b), f"frame {k}: filtered plane {i} differs"
        for fp, ref in ((whole, refs[0]), (parts, refs[1])):
            for d, s in zip(ref.planes(), fp.final_planes()):
                d.copy_(s)


@pytest.mark.parametrize("depth,width,height", [(8, 192, 128), (10, 320, 192), (8, 64, 64)])
def test_border_extension_of_a_pictures_planes_in_one_launch(depth, width, height):
    """x265hip_extend_border_planes (round 6): Y, Cb and Cr of one picture - each with its own geometry - extended by ONE launch must equal the three
    single-plane calls, which the closed-loop tests hold against the oracle (edge samples replicated into the margins, corners included)."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 1, depth=depth, seed=19)
    y, cb, cr = clip[0]
    a = P.DevicePicture(y, dev, cb, cr)
    b = P.DevicePicture(y, dev, cb, cr)
    for pic in (a, b):              # garbage in the margins: the extension must overwrite every margin sample
        for p in pic.planes():
            p.view(torch.uint8).fill_(0x5a)
    a2, b2 = P.DevicePicture(y, dev, cb, cr), P.DevicePicture(y, dev, cb, cr)
    # keep the interiors of the fresh pictures, wipe the margins only: copy the interior rows into the garbage-filled planes
    for dst, src in ((a, a2), (b, b2)):
        for k, (pd, ps) in enumerate(zip(dst.planes(), src.planes())):
            st, org, w, h = (dst.stride, dst.org, dst.w64, dst.h64) if k == 0 else (dst.stride_c, dst.org_c, dst.w64 // 2, dst.h64 // 2)
            for r in range(h):
                pd.view(-1)[org + r * st: org + r * st + w] = ps.view(-1)[org + r * st: org + r * st + w]
    S.extend_border(a.t, a)
    for i in range(2):
        S.extend_border(a.c[i], a, chroma=True)
    S.extend_border_picture(b.planes(), b)
    torch.cuda.synchronize()
    for pa, pb, pf in zip(a.planes(), b.planes(), a2.planes()):
        assert torch.equal(pa, pb)
        assert torch.equal(pa, pf)          # DevicePicture uploads planes whose margins the host already extended
