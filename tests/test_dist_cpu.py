"""World-size-2 gloo test of the multi-GPU seam (runs on CPU): frame-to-rank assignment and the
one-to-many reference-picture broadcast used by bench.py --gpus N."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fp = P.FrameParallel(rank, world)
    clip = F.synth_clip(128, 64, 6, seed=77)            # same clip on every rank
    planes = [torch.from_numpy(F.pad_plane(y)[0].copy()) for (y, _, _) in clip]
    ref = torch.zeros_like(planes[0])
    seen = []
    for step in range(3):
        mine = planes[fp.frame_index(step)]
        fp.exchange(ref, mine)
        seen.append(int(ref.to(torch.int64).sum()))
    expect = [int(planes[s * world + world - 1].to(torch.int64).sum()) for s in range(3)]
    out[rank] = (seen == expect, [fp.frame_index(s) for s in range(3)])
    dist.barrier()
    dist.destroy_process_group()


def test_reference_broadcast_two_ranks():
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] and out[1][0], "a rank did not receive the reference owner's picture"
    assert out[0][1] == [0, 2, 4] and out[1][1] == [1, 3, 5]      # frame-parallel dealing


def test_single_rank_is_a_plain_copy():
    P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
    fp = P.FrameParallel(0, 1)
    a, b = torch.zeros(8, dtype=torch.uint8), torch.arange(8, dtype=torch.uint8)
    fp.exchange(a, b)
    assert torch.equal(a, b)
