"""Pin the oracle: our C restatement (oracle/x265_oracle.c) against the REAL reference C
primitives compiled from /root/reference by oracle/Makefile (oracle/_ref/libx265ref*.so),
slot by slot, bit-exact - the reference TestBench contract (testbench.cpp:181-233).

Runs wherever oracle/_ref exists (this container; the prebuilt .so also travels to the
GPU box).  Skipped, not failed, when the reference build is absent."""
import pytest

import harness as H
import harness_host  # noqa: F401  (registers the handlers of the host-side slots)


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_oracle_matches_reference_all_slots(depth, repo_root):
    ref = H.load_reference(depth, repo_root)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    orc = H.load_oracle(depth, repo_root, host=True, entropy_from=ref)
    checked, fails = H.compare_tables(ref, orc, iters=3)
    assert checked >= 1860
    assert not fails, "\n".join(fails[:50])


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_null_pattern_matches_reference(depth, repo_root):
    """Exactly the slots the reference's cprim populates (setupCPrimitives + setupAliasPrimitives), all 2280 checked."""
    ref = H.load_reference(depth, repo_root)
    if ref is None:
        pytest.skip("oracle/_ref not built")
    orc = H.load_oracle(depth, repo_root, host=True)
    bad = [p for p in H.spec.SLOTS if bool(ref.ptr(p)) != bool(orc.ptr(p))]
    assert not bad, bad[:40]


def test_cabac_transition_table_matches_reference(repo_root):
    """The oracle derives the CABAC state transitions from the standard's transIdxLps; the reference ships a table."""
    import ctypes
    ref = H.load_reference(8, repo_root)
    if ref is None:
        pytest.skip("oracle/_ref not built")
    orc = H.load_oracle(8, repo_root, host=True)
    f = orc._owner[0].x265oracle_next_state_table_d8
    f.restype = ctypes.POINTER(ctypes.c_uint8 * 256)
    ours = bytes(f().contents)
    theirs = bytes((ctypes.c_uint8 * 256).in_dll(ref._owner, "_ZN4x26511g_nextStateE"))
    assert ours == theirs


def test_avx2_flavour_matches_plain(repo_root):
    """The -march=x86-64-v3 build used as CPU baseline computes the same bits."""
    import os
    if "avx2" not in open("/proc/cpuinfo").read():
        pytest.skip("host has no AVX2")
    a = H.load_oracle(8, repo_root)
    b = H.load_oracle(8, repo_root, avx2=True)
    checked, fails = H.compare_tables(a, b, iters=1, cases=("random",))
    assert not fails, "\n".join(fails[:20])
