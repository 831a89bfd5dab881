"""GPU parity, TABLE LAYER: the drop-in EncoderPrimitives table filled by x265hip_setup_primitives
vs the oracle table, slot by slot (the reference TestBench contract, testbench.cpp:181-233:
`if (opt.slot) check(ref.slot, opt.slot)`), through the real host-pointer signatures."""
import ctypes
import importlib

import pytest

import harness as H

pytestmark = pytest.mark.gpu

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
spec = H.spec


def load_hip_table(depth, base=None):
    """base: a Table to copy first (the host's own C table), then GPU slots are overwritten."""
    L = A.lib()
    mem = (ctypes.c_void_p * spec.TABLE_PTRS)()
    if base is not None:
        ctypes.memmove(mem, base.addr, spec.TABLE_BYTES)
    L.x265hip_setup_primitives.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = A.check(L.x265hip_setup_primitives(ctypes.byref(mem), spec.TABLE_BYTES, depth), "x265hip_setup_primitives")
    return spec.Table(ctypes.addressof(mem), depth, (L, mem)), n


@pytest.mark.parametrize("depth", [8, 10])
def test_table_slots_match_oracle(depth, repo_root):
    orc = H.load_oracle(depth, repo_root)
    hip, nset = load_hip_table(depth)
    assert nset > 100
    calls0 = A.lib().x265hip_table_calls()
    paths = [p for p in spec.SLOTS if hip.ptr(p)]
    assert len(paths) == nset
    # every GPU slot must exist in the reference-shaped table too (never fill a slot the reference leaves NULL)
    assert all(orc.ptr(p) for p in paths), [p for p in paths if not orc.ptr(p)][:10]
    checked, fails = H.compare_tables(orc, hip, paths=paths, iters=1)
    assert checked == len(paths)
    assert not fails, "\n".join(fails[:40])
    assert A.lib().x265hip_table_calls() > calls0, "stubs did not go through the GPU path"


def test_table_rejects_bad_arguments():
    L = A.lib()
    L.x265hip_setup_primitives.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    mem = (ctypes.c_void_p * spec.TABLE_PTRS)()
    assert L.x265hip_setup_primitives(ctypes.byref(mem), 100, 8) < 0
    assert L.x265hip_setup_primitives(ctypes.byref(mem), spec.TABLE_BYTES, 9) < 0
    assert L.x265hip_setup_primitives(None, spec.TABLE_BYTES, 8) < 0


@pytest.mark.parametrize("depth", [8, 10])
def test_whole_plane_weight_pp_on_a_fresh_thread(depth, repo_root):
    """The reference calls weight_pp on whole lowres planes (slicetype.cpp:821,956: stride x paddedLines,
    ~680 KB at 1080p, ~2.4 MB at 4K).  A fresh host thread starts with 1 MiB of staging, so the output
    allocation regrows (moves) the staging buffers after the job record was packed: the record's device
    address must be resolved after the last allocation (round-1 advisor finding)."""
    import threading
    import numpy as np
    orc = H.load_oracle(depth, repo_root)
    hip, _ = load_hip_table(depth)
    rng = np.random.default_rng(77 + depth)
    w, h, st = 1088, 700, 1120                      # 784 000 samples: > 512 KiB in, > 1 MiB with the output
    src = H.pixels(rng, "random", depth, w, h, st)
    corr = 14 - depth
    args = (37, (1 << (corr + 5)) & ~((1 << corr) - 1), corr + 6, -9)
    outs, errs = [], []

    def work(tab):
        try:
            d = H.out2d(w, h, st, H.pix_dtype(depth), 0x33)
            tab.fn("weight_pp")(src.p, d.p, st, w, h, *args)
            outs.append(d.data)
        except Exception as e:              # pragma: no cover
            errs.append(e)
    for tab in (orc, hip):
        t = threading.Thread(target=work, args=(tab,))
        t.start(); t.join()
    assert not errs and len(outs) == 2
    assert np.array_equal(outs[0], outs[1])
