"""The multi-GPU seam's C entries on ONE GPU: a communicator of one rank (x265hip_comm_unique_id / _init / _destroy) and a band published
over it (x265hip_recon_publish_rows, the one-to-many form: a broadcast over one rank moves nothing).  What it proves on a 1-GPU box: librccl
opens through the library's own dlopen, every entry the ring's AbiTransport calls resolves and returns, the band's slices are accepted by
RCCL (pointers, sizes, group calls) and the planes come back untouched.  The two-rank transfer itself runs in `bench.py --gpus N` at round
end; its protocol is covered by the gloo twin (tests/test_dist_cpu.py).  The work runs in a child process under a time limit: an RCCL
call that blocks must fail this test, not hang the box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
L = A.lib()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
L.x265hip_comm_unique_id.argtypes = [ctypes.c_void_p]
L.x265hip_comm_init.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
L.x265hip_comm_destroy.argtypes = [ctypes.c_void_p]
uid = (ctypes.c_uint8 * 128)()
A.check(L.x265hip_comm_unique_id(uid), "x265hip_comm_unique_id")
assert any(uid), "an all-zero id"
comm = ctypes.c_void_p()
A.check(L.x265hip_comm_init(ctypes.byref(comm), 1, bytes(uid), 0), "x265hip_comm_init")
assert comm.value
for depth in (8, 10):
    dt = torch.uint8 if depth == 8 else torch.int16
    H, st, my, sc, myc = 256, 448, 80, 256, 40
    g = torch.Generator().manual_seed(depth)
    planes = [torch.randint(0, 200, ((H + 2 * my) * st,), generator=g, dtype=torch.int32).to(dt).to(dev),
              torch.randint(0, 200, ((H // 2 + 2 * myc) * sc,), generator=g, dtype=torch.int32).to(dt).to(dev),
              torch.randint(0, 200, ((H // 2 + 2 * myc) * sc,), generator=g, dtype=torch.int32).to(dt).to(dev)]
    before = [p.clone() for p in planes]
    f = L.x265hip_recon_publish_rows
    f.argtypes = [ctypes.POINTER(A.ReconPublishParams), ctypes.c_void_p]
    s = torch.cuda.Stream(device=dev)
    for row0, rows in ((0, 1), (1, 2), (3, 1), (0, 4)):
        p = A.ReconPublishParams()
        p.comm, p.rank, p.root, p.peer, p.depth = comm, 0, 0, -1, depth
        for i in range(3):
            p.plane[i] = planes[i].data_ptr()
        p.stride, p.stride_c, p.margin_y, p.margin_y_c, p.height, p.ctu_row0, p.ctu_rows = st, sc, my, myc, H, row0, rows
        A.check(f(ctypes.byref(p), s.cuda_stream), "x265hip_recon_publish_rows")
    s.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(planes, before))
    # the argument checks still answer with a live communicator
    p.ctu_row0, p.ctu_rows = 3, 2
    assert f(ctypes.byref(p), s.cuda_stream) == -2
assert L.x265hip_comm_destroy(comm) == 0
print("RING-ABI-OK")
'''


def test_one_rank_communicator_and_a_published_band_through_the_c_abi():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "RING-ABI-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


CHILD_BCAST = r'''
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
P = importlib.import_module("x265-yuuki-asuna_amd.pipeline")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
H, st, my, sc, myc = 256, 448, 80, 256, 40
geom = (st, my, sc, myc)
T = P.AbiBcastTransport(0, 1, dev, 8, geom, H)
T.setup(dev)
g = torch.Generator().manual_seed(5)
planes = [torch.randint(0, 200, ((H + 2 * my) * st,), generator=g, dtype=torch.int32).to(torch.uint8).to(dev),
          torch.randint(0, 200, ((H // 2 + 2 * myc) * sc,), generator=g, dtype=torch.int32).to(torch.uint8).to(dev),
          torch.randint(0, 200, ((H // 2 + 2 * myc) * sc,), generator=g, dtype=torch.int32).to(torch.uint8).to(dev)]
before = [p.clone() for p in planes]
bands = [(0, 2), (2, 2)]
works = []
for b, (r0, rn) in enumerate(bands):
    rows = P.FrameParallelRing._rows(geom, r0, rn, b == 0, b == len(bands) - 1)
    works += T.send(planes, rows, (r0, rn), [])          # the root's side of the broadcast
    works += T.recv(planes, rows, (r0, rn), 0)           # (one rank: the same call with root = itself)
for w in works:
    w.wait()
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(planes, before))
T.close()
print("RING-BCAST-OK")
'''


def test_broadcast_transport_on_a_one_rank_communicator():
    """pipeline.AbiBcastTransport (X265HIP_RING_TRANSPORT=bcast in bench.py): ONE communicator over all ranks, a band = a group of ncclBroadcast calls rooted at the
    producer on one copy stream.  On a 1-GPU box: the communicator of one rank is built by the transport's own setup, bands are published as the ring would (root and
    receiver side), the planes come back untouched.  The N-rank protocol - who joins what in which order - is covered on gloo (tests/test_dist_cpu.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    r = subprocess.run([sys.executable, "-c", CHILD_BCAST], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "RING-BCAST-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
