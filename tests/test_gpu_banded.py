"""GPU parity of the banded frame pipeline (stages.BandedFramePipeline - the reference's `--slices` picture: every band of CTU rows is
searched, reconstructed and loop-filtered as a slice of its own while search and prediction read the whole reference picture) against
the oracle chain run band by band with the same sub-picture addressing, over a closed loop of frames."""
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


@pytest.mark.parametrize("depth,band_rows,par", [(8, 2, False), (8, 3, False), (10, 2, False), (8, 2, True), (10, 3, True)])
def test_banded_pipeline_equals_oracle_band_by_band(depth, band_rows, par):
    """par: the Cb / Cr chains of every band on their own HIP streams (bench.py's launch structure)."""
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    dev = torch.device("cuda:0")
    W, Hh, R, subme, level, qp = 256, 320, 12, 3, 2, 30 + 12 * (depth == 10)          # 5 CTU rows: bands of 2 + 2 + 1 / 3 + 2
    clip = F.synth_clip(W, Hh, 3, depth=depth, seed=71)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    bp = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, band_rows=band_rows, rng=R, subme=subme, level=level, qp=qp, want_surf=False,
                               deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True, parallel_planes=par)
    assert [n for _, n in bp.bands] == ([2, 2, 1] if band_rows == 2 else [3, 2])
    ref_dev = pics[0].like([p.clone() for p in pics[0].planes()])
    ref_host = None
    ctus_w = pics[0].w64 // 64
    dt = pics[0].host.dtype
    for k in (1, 2):
        got = {}

        def grab(b, row0, n, k=k):          # copy the band's stage outputs before the next band reuses the stage buffers
            torch.cuda.synchronize()
            pipe = bp.pipes[n]
            got[b] = {"me_best": pipe.ms.best.cpu().numpy().view(np.uint64).copy(), "subpel_mv": pipe.sp.out.cpu().numpy().reshape(-1, 2).copy(),
                      "levels": pipe.rc.levels.cpu().numpy().copy(), "num_sig": pipe.rc.num_sig.cpu().numpy().copy(),
                      "sao_params": pipe.sao.params.cpu().numpy().copy(), "levels_c0": pipe.rc_c[0].levels.cpu().numpy().copy(),
                      "sao_params_c1": pipe.sao_c[1].params.cpu().numpy().copy()}
        bp.run(pics[k], ref_dev, band_ready=grab)
        torch.cuda.synchronize()
        planes = [p.cpu().numpy().view(dt) for p in bp.final_planes()]
        y_plane = planes[0].reshape(pics[k].host.shape)
        next_host = [np.zeros_like(y_plane), np.zeros_like(planes[1]), np.zeros_like(planes[2])]
        sc = pics[k].stride_c
        for b, (row0, n) in enumerate(bp.bands):
            _, exp = B.oracle_chain(F, clip, R, subme, level, qp, depth, n * ctus_w, 4, False, ref_planes=ref_host, cur_index=k, band=(row0, n))
            for key, val in got[b].items():
                e = np.asarray(exp[key]).reshape(-1)
                assert np.array_equal(val.reshape(-1).astype(np.int64), e.astype(np.int64)), f"frame {k} band {b}: {key} differs"
            first, last = b == 0, b == len(bp.bands) - 1
            r0 = F.MARGIN_Y + row0 * 64 - F.MARGIN_Y * first
            r1 = F.MARGIN_Y + (row0 + n) * 64 + F.MARGIN_Y * last
            assert np.array_equal(y_plane[r0:r1], exp["recon"]), f"frame {k} band {b}: filtered luma rows differ"
            next_host[0][r0:r1] = exp["recon"]
            c0 = F.CHROMA_MARGIN_Y + row0 * 32 - F.CHROMA_MARGIN_Y * first
            c1 = F.CHROMA_MARGIN_Y + (row0 + n) * 32 + F.CHROMA_MARGIN_Y * last
            for i in range(2):
                gc = planes[1 + i].reshape(-1, sc)[c0:c1]
                assert np.array_equal(gc, exp["recon_c%d" % i]), f"frame {k} band {b}: filtered chroma plane {i} differs"
                next_host[1 + i].reshape(-1, sc)[c0:c1] = exp["recon_c%d" % i]
        # the filtered picture becomes the next frame's reference on both sides
        ref_host = (next_host[0], next_host[1].reshape(-1), next_host[2].reshape(-1))
        ref_dev = pics[k].like([p.clone() for p in bp.final_planes()])


@pytest.mark.parametrize("depth,packed", [(8, True), (10, False)])
def test_banded_graph_replay_and_band_streams_equal_launch_by_launch(depth, packed):
    """graphs=True: every band's launches recorded once per (band, source, reference) as a HIP graph and replayed - a closed loop of six
    frames (the filtered picture is the next reference, buffers stay in place) gives the planes of the launch-by-launch pipeline."""
    import torch
    dev = torch.device("cuda:0")
    W, Hh, R, subme, level, qp = 256, 320, 12, 3, 2, 30 + 12 * (depth == 10)
    clip = F.synth_clip(W, Hh, 3, depth=depth, seed=72)
    pics = [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]
    kw = dict(band_rows=2, rng=R, subme=subme, level=level, qp=qp, want_surf=True, packed=packed, deblock=True, sao=True, chroma=True, sao_apply=True,
              sign_hide=True, lookahead=(W, Hh))
    eager = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, **kw)
    graph = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, graphs=True, **kw)
    # bands alternating between two HIP streams (each with its own stage buffers): the search of band b + 1 overlaps the filters of band b
    two = S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, streams=2, **kw)
    refs = [pics[0].like([p.clone() for p in pics[0].planes()]) for _ in range(3)]
    for pc in pics[1:]:
        graph.capture(pc, refs[1])
    assert len(graph.graphs) == 2 * len(graph.bands)              # capture() visits every band twice: a band that warmed its stage buffers up is recorded on the second visit
    recorded = len(graph.graphs)
    for f in range(6):
        cur = pics[1 + f % 2]
        for bp, ref in ((eager, refs[0]), (graph, refs[1]), (two, refs[2])):
            bp.run(cur, ref)
            for d, s in zip(ref.planes(), bp.final_planes()):
                d.copy_(s)
        torch.cuda.synchronize()
        for i, (a, b, c) in enumerate(zip(refs[0].planes(), refs[1].planes(), refs[2].planes())):
            assert torch.equal(a, b), f"frame {f}: plane {i} of the graph replay differs"
            assert torch.equal(a, c), f"frame {f}: plane {i} of the two-stream bands differs"
        assert torch.equal(eager.la.intra_cost, graph.la.intra_cost)
    assert len(graph.graphs) == recorded                          # the loop only replayed
