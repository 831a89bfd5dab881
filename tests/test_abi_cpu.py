"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/x265hip.h declares (no compute calls without a GPU), and the product never touches the oracle."""
import ctypes
import importlib
import os
import sys
import re

import numpy as np

A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")


def test_library_exports_every_declared_symbol():
    L = A.lib()
    for name in A.exported_symbols():
        assert hasattr(L, name), f"libx265hip.so does not export {name}"
    assert b"gfx950" in L.x265hip_version()


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    L = A.lib()
    out = (ctypes.c_uint64 * 1)()
    buf = (ctypes.c_uint8 * 4096)()
    rc = L.x265hip_pixelcmp_batch(0, 8, 8, 8, buf, 8, None, 64, buf, 8, None, 64, 1, out, None)
    assert rc < 0 and b"no HIP device" in L.x265hip_last_error() or rc < 0


def test_product_never_references_oracle(repo_root):
    """The judge's rule: product code must not import / link / execute anything under oracle/."""
    pkg = os.path.join(repo_root, "x265-yuuki-asuna_amd")
    bad = []
    for dp, dn, fn in os.walk(pkg):
        if "_obj" in dp or "__pycache__" in dp:
            continue
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"oracle_api|libx265oracle|x265oracle_|oracle/_build|oracle/_ref|from oracle|import oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    # the shared object must not depend on the oracle library either
    import subprocess
    deps = subprocess.run(["ldd", A.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps


def test_ctypes_structures_match_the_c_header(repo_root, tmp_path):
    """Every ctypes mirror in hipabi.py has the size and the field offsets of its struct in include/x265hip.h, as gcc lays it out:
    a field added on one side only would silently shift everything behind it."""
    import ctypes
    import importlib
    import subprocess
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    S = importlib.import_module("x265-yuuki-asuna_amd.stages")
    pairs = {"x265hip_me_params": A.MEParams, "x265hip_subpel_params": A.SubpelParams, "x265hip_lowres_init_params": A.LowresInitParams,
             "x265hip_lowres_intra_params": A.LowresIntraParams, "x265hip_lowres_cost_pair": A.LowresCostPair,
             "x265hip_lowres_cost_params": A.LowresCostParams, "x265hip_me_search_job": A.MESearchJob, "x265hip_me_search_params": A.MESearchParams,
             "x265hip_sea_integral_params": A.SeaIntegralParams, "x265hip_deblock_bs_params": A.DeblockBsParams,
             "x265hip_deblock_chroma_params": A.DeblockChromaParams, "x265hip_deblock_params": A.DeblockParams,
             "x265hip_aq_energy_params": A.AqEnergyParams, "x265hip_aq_offsets_params": A.AqOffsetsParams, "x265hip_aq_frame_host_params": A.AqFrameHostParams,
             "x265hip_weight_analyse_ref": A.WeightAnalyseRef, "x265hip_weight_analyse_host_params": A.WeightAnalyseHostParams,
             "x265hip_cutree_propagate_params": A.CuTreePropagateParams, "x265hip_cutree_finish_params": A.CuTreeFinishParams, "x265hip_frame_cost_recalculate_params": A.FrameCostRecalculateParams,
             "x265hip_lowres_weight_cost_params": A.LowresWeightCostParams, "x265hip_lowres_weight_apply_params": A.LowresWeightApplyParams,
             "x265hip_sao_stats_params": A.SaoStatsParams, "x265hip_sao_apply_params": A.SaoApplyParams, "x265hip_plane": A.Plane,
             "x265hip_intra_recon_params": A.IntraReconParams, "x265hip_tu_tables": A.TuTablesRec, "x265hip_phase_planes_params": A.PhasePlanesParams,
             "x265hip_aq_hevc_params": A.AqHevcParams, "x265hip_aq_hevc_offsets_params": A.AqHevcOffsetsParams,
             "x265hip_cutree_finish_hevc_params": A.CuTreeFinishHevcParams}
    sys.path.insert(0, repo_root)
    from tools import seam_driver as SD          # the consumer-layer records the seam tools mirror
    pairs.update({"x265hip_me_cache_params": SD.CacheParams, "x265hip_me_cache_stats_t": SD.CacheStats,
                  "x265hip_phase_cache_params": SD.PhaseCacheParams, "x265hip_phase_cache_stats_t": SD.PhaseCacheStats})
    for extra, cname in (("ReconParams", "x265hip_recon_params"), ("ReconBiParams", "x265hip_recon_bi_params"), ("PredWeight", "x265hip_pred_weight")):
        cls = getattr(A, extra, None) or getattr(S, extra, None)
        if cls is not None and isinstance(cls, type) and issubclass(cls, ctypes.Structure):
            pairs[cname] = cls
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "x265hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} . %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(repo_root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, fname, val = line.split()
        cls = pairs[cname]
        expect = ctypes.sizeof(cls) if fname == "." else getattr(cls, fname).offset
        assert int(val) == expect, f"{cname}.{fname}: C says {val}, ctypes says {expect}"
        seen += 1
    assert seen > 150


def test_library_has_no_unresolved_symbols_of_its_own():
    """Every x265hip / x265hip:: symbol the library references must be defined inside it (a shared library links with undefined symbols)."""
    import subprocess
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x265-yuuki-asuna_amd", "libx265hip.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    bad = [l for l in out.splitlines() if "x265hip" in l]
    assert not bad, bad


def test_recon_publish_rows_validates_before_touching_a_device():
    """The multi-GPU seam's C entry rejects bad geometry without a device or RCCL (argument checks come first)."""
    import ctypes

    class PP(ctypes.Structure):
        _fields_ = [("comm", ctypes.c_void_p), ("rank", ctypes.c_int), ("root", ctypes.c_int), ("peer", ctypes.c_int), ("depth", ctypes.c_int),
                    ("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_ssize_t), ("stride_c", ctypes.c_ssize_t), ("margin_y", ctypes.c_int),
                    ("margin_y_c", ctypes.c_int), ("height", ctypes.c_int), ("ctu_row0", ctypes.c_int), ("ctu_rows", ctypes.c_int)]
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    f = A.lib().x265hip_recon_publish_rows
    f.argtypes = [ctypes.POINTER(PP), ctypes.c_void_p]
    p = PP()
    assert f(ctypes.byref(p), None) == -2                       # NULL communicator
    p.comm, p.depth, p.height, p.ctu_row0, p.ctu_rows, p.stride, p.margin_y = 1, 8, 128, 1, 2, 320, 80
    p.plane[0] = 4096
    assert f(ctypes.byref(p), None) == -2                       # rows 1..3 of a 2-row picture
    assert b"rows" in A.lib().x265hip_last_error()


def test_phase_plane_entries_validate_before_touching_a_device():
    """x265hip_phase_planes / x265hip_phase_cache_create reject bad geometry without a device."""
    import ctypes
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    L = A.lib()
    f = L.x265hip_phase_planes
    f.argtypes = [ctypes.POINTER(A.PhasePlanesParams), ctypes.c_void_p]
    assert f(None, None) == -2
    assert f(ctypes.byref(A.PhasePlanesParams(8, 0, 4096, 8192, 130, 64)), None) == -2          # stride not a multiple of 4
    assert f(ctypes.byref(A.PhasePlanesParams(8, 0, 4096, 8192, 128, 62)), None) == -2          # rows not a multiple of 4
    assert f(ctypes.byref(A.PhasePlanesParams(9, 0, 4096, 8192, 128, 64)), None) == -2          # depth
    assert f(ctypes.byref(A.PhasePlanesParams(8, 1, 4096, 8192, 128, 12)), None) == -2          # too few rows to produce any
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import seam_driver as SD
    g = L.x265hip_phase_cache_create
    g.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(SD.PhaseCacheParams)]
    h = ctypes.c_void_p()
    assert g(ctypes.byref(h), ctypes.byref(SD.PhaseCacheParams(8, 130, 64, 0, 0, 2))) == -2 and not h
    assert g(ctypes.byref(h), ctypes.byref(SD.PhaseCacheParams(8, 128, 64, 64, 30, 2))) == -2   # chroma rows not a multiple of 4
    assert g(ctypes.byref(h), ctypes.byref(SD.PhaseCacheParams(8, 128, 64, 0, 0, 0))) == -2     # no slots


def test_sao_planes_validates_before_touching_a_device():
    """x265hip_sao_planes rejects a bad plane count / mixed bit depths / missing buffers without a device."""
    import ctypes
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    L = A.lib()
    f = L.x265hip_sao_planes
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    st = (A.SaoStatsParams * 3)()
    assert f(0, ctypes.cast(st, ctypes.c_void_p), None, None) == -2
    assert f(4, ctypes.cast(st, ctypes.c_void_p), None, None) == -2
    assert f(1, None, None, None) == -2
    assert f(1, ctypes.cast(st, ctypes.c_void_p), None, None) == -2          # empty record: no planes, no geometry
    for i in range(2):
        st[i].depth, st[i].fenc, st[i].fenc_stride, st[i].rec, st[i].rec_stride = 8 + 2 * i, 4096, 256, 8192, 256
        st[i].width, st[i].height, st[i].count, st[i].offset_org = 128, 64, 12288, 16384
    assert f(2, ctypes.cast(st, ctypes.c_void_p), None, None) == -2          # 8-bit and 10-bit planes in one call


def test_chroma_pair_validates_before_touching_a_device():
    """x265hip_inter_recon_chroma_pair needs two complete records of one geometry (checked after the device: only the NULL case here)."""
    import ctypes
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    f = A.lib().x265hip_inter_recon_chroma_pair
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert f(None, None, None) == -2


def test_hevc_aq_host_side_matches_the_oracle_without_a_device():
    """x265hip_aq_hevc_offsets is host code: the double-precision half of --hevc-aq from the oracle's quadrant sums equals the oracle's
    (reference-pinned) restatement bit for bit, incl. clipped partitions; the device entry refuses bad geometry before touching a device."""
    import ctypes
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_api as O
    # (the arbitrary range matters: a compiler that turns pow(2.0, x) into exp2(x) is one ulp off for most x, and exact for the round ones)
    for depth, width, height, qg, rng in ((8, 250, 138, 16, 1.0), (10, 232, 120, 8, 3.0), (8, 208, 144, 64, 6.0), (10, 304, 206, 16, 4.0762005658023845),
                                          (8, 128, 72, 32, 1.7320508075688772), (12, 96, 64, 8, 5.123456789)):
        y = F.synth_clip(width, height, 1, depth=depth, seed=7)[0][0]
        yp, stride, org, _, _ = F.pad_plane(y)
        parts, act, qp, avg, inv, _, _ = O.aq_hevc_frame(depth, yp, stride, org, width, height, qg_size=qg, qp_adaptation_range=rng)
        at = 0
        for d in range(4):
            if not parts[d]:
                continue
            part = 64 >> d
            sums = O.aq_hevc_quadrants(depth, yp, stride, org, width, height, part)
            a, q, g, iv = A.aq_hevc_offsets(width, height, part, rng, sums)
            assert np.array_equal(a, act[at:at + parts[d]]) and np.array_equal(q, qp[at:at + parts[d]]) and g == avg[d]
            if d == max(k for k in range(4) if parts[k]):
                assert np.array_equal(iv, inv)
            at += parts[d]
    f = A.lib().x265hip_aq_hevc_quadrants
    f.argtypes = [ctypes.POINTER(A.AqHevcParams), ctypes.c_void_p]
    assert f(None, None) == -2
    assert f(ctypes.byref(A.AqHevcParams(8, 4096, 256, 128, 64, 24, 8192)), None) == -2        # partition size
    assert f(ctypes.byref(A.AqHevcParams(8, 4096, 256, 127, 64, 16, 8192)), None) == -2        # odd width


def test_bit_cost_table_is_required_before_the_estimators_that_price_bins_run():
    """x265hip_coeff_batch(COST_COEFF_NXN / COST_C1C2) without x265hip_set_entropy_bits must fail with a message, not price bins with a table of its own: the
    product holds no copy of the encoder's CABAC bit costs (the source says so: no 128-entry constant table in csrc/frame_coeff_kernels.hip)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "x265-yuuki-asuna_amd", "csrc", "frame_coeff_kernels.hip")).read()
    assert "were not handed in (x265hip_set_entropy_bits)" in src
    assert not re.search(r"0x[0-9A-Fa-f]{8}\s*,\s*0x[0-9A-Fa-f]{8}\s*,\s*0x[0-9A-Fa-f]{8}", src), "a packed constant table in the product"
