"""GPU check of the N > 1 path on ONE device: two / three processes share cuda:0, frame f is encoded by rank f % world with
stages.BandedFramePipeline (bands alternating between HIP streams) and searches frame f - 1, whose bands arrive through
pipeline.FrameParallelRing (gloo, staged through host memory - the transfers order themselves against the band streams exactly as RCCL's
do).  Every frame's filtered planes must equal those of one process encoding the same frames one after the other on one stream."""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, HH, R, SUBME, LEVEL, QP, STEPS = 256, 448, 12, 3, 2, 30, 3          # 7 CTU rows: bands of 2 + 2 + 2 + 1


def _setup(depth, streams):
    import torch
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    dev = torch.device("cuda:0")
    return F, P, S, dev


def _clip(F, P, dev, depth, nframes):
    clip = F.synth_clip(W, HH, nframes + 1, depth=depth, seed=83)
    return [P.DevicePicture(y, dev, u, v) for (y, u, v) in clip]


def _pipeline(S, pics, depth, dev, streams, tiled=False):
    return S.BandedFramePipeline(pics[0].w64, pics[0].h64, depth, dev, band_rows=2, rng=R, subme=SUBME, level=LEVEL, qp=QP + 12 * (depth == 10),
                                 want_surf=True, packed=("t" if tiled else True) if depth == 8 else False, deblock=True, sao=True, chroma=True, sao_apply=True, sign_hide=True,
                                 lookahead=(W, HH), streams=streams)


def _digest(planes):
    import torch
    torch.cuda.synchronize()
    h = hashlib.md5()
    for p in planes:
        h.update(p.cpu().numpy().tobytes())
    return h.hexdigest()


def _worker(rank, world, port, depth, tiled, out, gop=0):
    import torch
    import torch.distributed as dist
    F, P, S, dev = _setup(depth, 3)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = STEPS * world
    pics = _clip(F, P, dev, depth, total)
    bp = _pipeline(S, pics, depth, dev, 3, tiled)
    ring = P.FrameParallelRing(rank, world, bp.bands, lag_rows_luma=R + 16, stage_through_host=True, gop=gop)
    ring.make_groups()
    geom = (pics[0].stride, F.MARGIN_Y, pics[0].stride_c, F.CHROMA_MARGIN_Y)
    ref = pics[0].like([p.clone() for p in pics[0].planes()])
    got = {}
    for step in range(STEPS):
        f = ring.frame_index(step)
        cur = pics[1 + f]
        ring.finish()
        bp.begin_frame(cur)
        ring.run_frame(step, geom, ref.planes(), bp.final_planes(), lambda b, row0, n: bp.run_band(b, cur, ref), total_frames=total,
                       band_context=bp.band_context)
        bp.end_frame()
        ring.finish()                        # the digest below reads the planes the sends are reading: both only read
        got[f] = _digest(bp.final_planes())
    ring.finish()
    out[rank] = got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("depth,world,tiled,gop", [(8, 2, False, 0), (8, 3, False, 0), (10, 2, False, 0), (8, 2, True, 0), (8, 2, False, 3), (8, 3, False, 2)])
def test_ring_on_a_shared_device_equals_one_process(depth, world, tiled, gop):
    """tiled: the ranks search with the record-per-lane kernel (chunk-major surfaces), the one-process encode with the row-walking one.
    gop: mini-GOPs (round 6) - every picture reads the newest multiple of gop before it; the one-process encode keeps that anchor as its reference."""
    import torch
    import torch.multiprocessing as mp
    F, P, S, dev = _setup(depth, 1)
    total = STEPS * world
    pics = _clip(F, P, dev, depth, total)
    bp = _pipeline(S, pics, depth, dev, 1)
    ref = pics[0].like([p.clone() for p in pics[0].planes()])
    expect = {}
    for f in range(total):
        bp.run(pics[1 + f], ref)
        if not gop or f % gop == 0:                      # chain: every picture is the next one's reference; mini-GOPs: the anchors only
            for d, s in zip(ref.planes(), bp.final_planes()):
                d.copy_(s)
        expect[f] = _digest(bp.final_planes())
    assert len(set(expect.values())) == total
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, depth, tiled, out, gop), nprocs=world, join=True)
    got = {}
    for r in range(world):
        assert sorted(out[r]) == [s * world + r for s in range(STEPS)]
        got.update(out[r])
    bad = [f for f in range(total) if got[f] != expect[f]]
    assert not bad, f"frames {bad} of the {world}-rank ring differ from the one-process encode"
