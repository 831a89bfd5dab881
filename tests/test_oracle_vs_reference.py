"""Pin the oracle: our C restatement (oracle/x265_oracle.c) against the REAL reference C
primitives compiled from /root/reference by oracle/Makefile (oracle/_ref/libx265ref*.so),
slot by slot, bit-exact - the reference TestBench contract (testbench.cpp:181-233).

Runs wherever oracle/_ref exists (this container; the prebuilt .so also travels to the
GPU box).  Skipped, not failed, when the reference build is absent."""
import pytest

import harness as H


@pytest.mark.parametrize("depth", [8, 10])
def test_oracle_matches_reference_all_slots(depth, repo_root):
    ref = H.load_reference(depth, repo_root)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    orc = H.load_oracle(depth, repo_root)
    checked, fails = H.compare_tables(ref, orc, iters=3)
    assert checked >= 1800
    assert not fails, "\n".join(fails[:50])


@pytest.mark.parametrize("depth", [8, 10])
def test_null_pattern_matches_reference(depth, repo_root):
    """Same slots populated as the reference's cprim (setupCPrimitives + setupAliasPrimitives),
    except rows a9/a16 of SURVEY section 8 which the restatement does not cover yet."""
    ref = H.load_reference(depth, repo_root)
    if ref is None:
        pytest.skip("oracle/_ref not built")
    orc = H.load_oracle(depth, repo_root)
    uncovered = {"nonPsyRdoQuant", "psyRdoQuant", "psyRdoQuant_1p", "psyRdoQuant_2p", "ssim_4x4x2_core",
                 "ssim_end_4", "frameInitLowres", "frameInitLowerRes", "propagateCost", "fix8Unpack",
                 "fix8Pack", "planecopy_cp", "planecopy_sp", "planecopy_sp_shl", "planecopy_pp_shr",
                 "planeClipAndMax", "scanPosLast", "findPosFirstLast", "costCoeffNxN", "costCoeffRemain",
                 "costC1C2Flag"}
    bad = [p for p in H.spec.SLOTS
           if bool(ref.ptr(p)) != bool(orc.ptr(p)) and H.field_of(p) not in uncovered]
    assert not bad, bad[:40]


def test_avx2_flavour_matches_plain(repo_root):
    """The -march=x86-64-v3 build used as CPU baseline computes the same bits."""
    import os
    if "avx2" not in open("/proc/cpuinfo").read():
        pytest.skip("host has no AVX2")
    a = H.load_oracle(8, repo_root)
    b = H.load_oracle(8, repo_root, avx2=True)
    checked, fails = H.compare_tables(a, b, iters=1, cases=("random",))
    assert not fails, "\n".join(fails[:20])
