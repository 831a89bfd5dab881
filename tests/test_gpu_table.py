"""GPU parity, TABLE LAYER: the drop-in EncoderPrimitives table filled by x265hip_setup_primitives
vs the oracle table, slot by slot (the reference TestBench contract, testbench.cpp:181-233:
`if (opt.slot) check(ref.slot, opt.slot)`), through the real host-pointer signatures."""
import ctypes
import importlib

import pytest

import harness as H
import harness_host  # noqa: F401  (registers the handlers of rows a9 / a16)

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


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_table_slots_match_oracle(depth, repo_root):
    orc = H.load_oracle(depth, repo_root, host=True)          # rows a9 / a16 included: the table is complete
    A.set_entropy_bits(H.host_tables(repo_root)["entropy_bits"])      # the host's CABAC bit costs (costCoeffNxN, costC1C2Flag)
    hip, nset = load_hip_table(depth)
    assert nset > 100
    assert all(hip.ptr(p) for p in spec.SLOTS if orc.ptr(p)), [p for p in spec.SLOTS if orc.ptr(p) and not hip.ptr(p)][:10]   # no slot is left to the host
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


def test_error_policy_restore_host_hands_the_slots_back(repo_root):
    """X265HIP_ON_ERROR_RESTORE_HOST: a failing stub (test hook: the 3rd call) is reported, that call and every later one are answered
    by the function the HOST had in the slot (here a sentinel table whose sad returns a recognisable value), nothing aborts; the
    default policy stays 'abort'."""
    import numpy as np
    L = A.lib()
    L.x265hip_table_failures.restype = ctypes.c_uint64
    L.x265hip_table_inject_failure.argtypes = [ctypes.c_long]
    orc = H.load_oracle(8, repo_root)
    hip, _ = load_hip_table(8, base=orc)                      # the host's own table first, GPU slots on top
    rng = np.random.default_rng(5)
    a, b = H.pixels(rng, "random", 8, 16, 16, 64), H.pixels(rng, "random", 8, 16, 16, 48)
    want = orc.fn("pu[2].sad")(a.p, a.stride, b.p, b.stride)
    f = hip.fn("pu[2].sad")
    assert f(a.p, a.stride, b.p, b.stride) == want            # through the GPU
    try:
        assert L.x265hip_set_error_policy(1) == 0
        fails0, calls0 = L.x265hip_table_failures(), L.x265hip_table_calls()
        L.x265hip_table_inject_failure(3)
        got = [f(a.p, a.stride, b.p, b.stride) for _ in range(6)]
        assert got == [want] * 6                              # identical answers: GPU, GPU, host (failed call), host, host, host
        assert L.x265hip_table_failures() == fails0 + 1
        assert L.x265hip_table_calls() == calls0 + 3          # calls 4..6 never reached a stub
        sx = hip.fn("pu[2].sad_x3")                           # every other GPU-backed slot is handed back as well
        res = (ctypes.c_int32 * 3)()
        sx(a.p, b.p, b.p + 1, b.p + 2, b.stride, res)
        assert res[0] == want and L.x265hip_table_calls() == calls0 + 3
    finally:
        L.x265hip_table_inject_failure(0)                     # clears the failed state
        L.x265hip_set_error_policy(0)
    assert f(a.p, a.stride, b.p, b.stride) == want and L.x265hip_table_calls() > calls0 + 3
    assert L.x265hip_set_error_policy(7) < 0


def test_restore_host_answers_every_slot_with_that_slots_own_host_function(repo_root):
    """One stub serves many slots ([0]/[1] pairs, dct / standard_dct, intra_pred[2..34], cu[1..4].normFact, the chroma tables'
    aliases of luma sizes); an asm-enabled host has a DIFFERENT function in each.  After a failure every GPU-backed slot must be
    answered by the function the host had in exactly that slot (round-2 advisor finding: the saved pointer was per stub).  The host
    table here holds one distinguishable dummy per slot."""
    L = A.lib()
    L.x265hip_table_inject_failure.argtypes = [ctypes.c_long]
    depth = 8
    seen, keep = [], []
    mem = (ctypes.c_void_p * spec.TABLE_PTRS)()

    def dummy(idx, proto, has_ret):
        def f(*_a):
            seen.append(idx)
            return (idx & 0x7fff) if has_ret else None
        cb = proto(f)
        keep.append(cb)
        return ctypes.cast(cb, ctypes.c_void_p).value
    for path, (td, idx) in spec.SLOTS.items():
        mem[idx] = dummy(idx, spec.prototype(td, depth), spec.TYPEDEFS[td][0].strip() != "void")
    host = spec.Table(ctypes.addressof(mem), depth, keep)
    hip, nset = load_hip_table(depth, base=host)
    gpu_paths = [p for p in spec.SLOTS if hip.ptr(p) != host.ptr(p)]
    assert len(gpu_paths) == nset > 1800
    try:
        assert L.x265hip_set_error_policy(1) == 0
        L.x265hip_table_inject_failure(1)                      # the very next stub call fails before it touches an operand
        for path in gpu_paths:
            td, idx = spec.SLOTS[path]
            ret, args = spec.TYPEDEFS[td]
            seen.clear()
            got = hip.fn(path)(*[None if a.strip().endswith("*") else 0 for a in args])
            assert seen == [idx], f"{path}: answered by the host function of slot {seen} instead of its own ({idx})"
            if ret.strip() != "void":
                assert got == (idx & 0x7fff), path
    finally:
        L.x265hip_table_inject_failure(0)
        L.x265hip_set_error_policy(0)
