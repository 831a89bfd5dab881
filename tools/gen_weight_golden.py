#!/usr/bin/env python3
"""tests/golden/weight_analyse_d{8,10}.npz: inputs of the frame encoder's weightAnalyse (encoder/weightPrediction.cpp:222-497) as the REAL encoder presented them on
fading clips, with the weights the REFERENCE's own function chose - generated here (needs oracle/_ref, i.e. /root/reference), consumed by
tests/test_golden.py (the oracle's restatement) and tests/test_gpu_weight_analyse.py (x265hip_weight_analyse_host) on boxes without the reference.

  python tools/gen_weight_golden.py

How: real encodes of 160x128 fades through binding/x265hip_x265_binding.cpp's weightAnalyse seam with verify on and X265REF_WA_DUMP set - the seam writes what the
provider was handed and what x265's weightAnalyse then left in the slice (binding/x265hip_x265_binding.cpp, wa_dump_arr).  A few slices per depth are kept: weighted P slices,
a B slice with two lists, a slice that keeps weight 1."""
import glob
import importlib
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_dump(path):
    out, raw = {}, open(path, "rb").read()
    o = 0
    while o < len(raw):
        n, = struct.unpack_from("<I", raw, o); o += 4
        name = raw[o:o + n].decode(); o += n
        elem, count = struct.unpack_from("<IQ", raw, o); o += 12
        dt = {1: np.uint8, 2: np.uint16, 4: np.int32, 8: np.uint64}[elem]
        out[name] = np.frombuffer(raw, dt, count, o).copy(); o += elem * count
    return out


def main():
    from tools import encoder_bench as EB, seam_driver as SD
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    w, h, n = 160, 128, 10
    for depth in (8, 10):
        kept = []
        for fade, extra in (((1.0, 0.35), [("bframes", "0")]), ((0.4, 1.0), [("weightb", None), ("bframes", "3")]), ((1.0, 1.0), [("bframes", "1")])):
            with tempfile.TemporaryDirectory() as d:
                os.environ["X265REF_WA_DUMP"] = d
                clip = F.synth_clip(w, h, n, depth=depth, seed=41, fade=fade)
                yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
                opts = [("pools", "1"), ("frame-threads", "1"), ("crf", "24")] + extra
                lib, filler, report, close, prov = SD.install(depth, w, h, provider="oracle", rng=4, slots=8, min_pu=128, verify=True, streamed=True, min_level=1,
                                                              weight_analyse="oracle", aq="oracle")
                try:
                    EB.encode(lib, yuv, w, h, n, "medium", opts, filler)
                    rep = report()
                finally:
                    close()
                    os.environ.pop("X265REF_WA_DUMP", None)
                assert rep["weight_analyse_seam"]["verify_mismatches"] == 0, rep["weight_analyse_seam"]
                recs = [read_dump(p) for p in sorted(glob.glob(os.path.join(d, "wa_*.bin")))]
            # per clip: the first weighted slice, the first two-list slice, the first unweighted one
            pick = {}
            for r in recs:
                nl = int(r["geo"][11])
                weighted = bool(r["ref0_expected"].reshape(3, 4)[0, 0])
                key = ("two" if nl == 2 else "one", weighted, r["ref0_mvs"].size > 0)
                pick.setdefault(key, r)
            kept += list(pick.values())
        arrays = {"count": np.array([len(kept)], np.int32)}
        for i, r in enumerate(kept):
            for k, v in r.items():
                arrays[f"s{i}_{k}"] = v
        path = os.path.join(ROOT, "tests", "golden", f"weight_analyse_d{depth}.npz")
        np.savez_compressed(path, **arrays)
        print(path, len(kept), "slices", os.path.getsize(path), "bytes;",
              [(int(r["geo"][11]), r["ref0_expected"].reshape(3, 4)[:, 0].tolist(), r["ref0_mvs"].size > 0) for r in kept])


def main_aq():
    """tests/golden/aq_frame_d{8,10}.npz the same way: the source picture x265's calcAdaptiveQuantFrame (slicetype.cpp:444) was handed and the Lowres arrays IT filled
    (X265REF_AQ_DUMP in the AQ seam), for AQ modes 1 - 3 and both quantisation-group sizes."""
    from tools import encoder_bench as EB, seam_driver as SD
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    w, h, n = 160, 128, 2
    for depth in (8, 10):
        kept = []
        for extra in ([("aq-mode", "1")], [("aq-mode", "2")], [("aq-mode", "3"), ("aq-strength", "1.4")], [("aq-mode", "2"), ("qg-size", "8")], [("aq-mode", "2"), ("no-weightp", None)]):
            with tempfile.TemporaryDirectory() as d:
                os.environ["X265REF_AQ_DUMP"] = d
                clip = F.synth_clip(w, h, n, depth=depth, seed=43)
                yuv = np.concatenate([np.concatenate([p.reshape(-1) for p in fr]) for fr in clip])
                opts = [("pools", "1"), ("frame-threads", "1"), ("crf", "24")] + extra
                lib, filler, report, close, prov = SD.install(depth, w, h, provider="oracle", rng=4, slots=8, min_pu=128, verify=True, streamed=True, min_level=1, aq="oracle")
                try:
                    EB.encode(lib, yuv, w, h, n, "medium", opts, filler)
                    rep = report()
                finally:
                    close()
                    os.environ.pop("X265REF_AQ_DUMP", None)
                assert rep["aq_seam"]["verify_mismatches"] == 0 and rep["aq_seam"]["pictures_served"] == n, rep["aq_seam"]
                kept.append(read_dump(sorted(glob.glob(os.path.join(d, "aq_*.bin")))[0]))
        arrays = {"count": np.array([len(kept)], np.int32)}
        for i, r in enumerate(kept):
            for k, v in r.items():
                arrays[f"s{i}_{k}"] = v.view(np.float64) if k in ("strength", "qp_aq_offset", "qp_cutree_offset") else v
        path = os.path.join(ROOT, "tests", "golden", f"aq_frame_d{depth}.npz")
        np.savez_compressed(path, **arrays)
        print(path, len(kept), "pictures", os.path.getsize(path), "bytes;", [(int(r["geo"][9]), int(r["geo"][10]), int(r["geo"][11])) for r in kept])


if __name__ == "__main__":
    main()
    main_aq()
