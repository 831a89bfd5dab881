"""Host-side inputs of the device stages that price bits (x265hip_sao_rdo): the tables an encoder owns and hands over - the library never
recomputes them - and the two context initialisations the HEVC standard defines.

  * `load(path)`: entropy_bits / lambda2_tab / chroma_scale as the host built them (tests/golden/host_tables.json is a dump of the
    reference build's own tables made by tools/gen_host_tables.py; a real host passes its g_entropyBits / x265_lambda2_tab directly).
  * `cabac_init_state(qp, init_value)`: ITU-T H.265 9.3.2.2 (the reference's sbacInit, entropy.cpp:1297-1308) -> the context byte
    pStateIdx << 1 | valMps.
  * `sao_contexts(slice_type, slice_qp)`: the initial states of sao_merge_left/up_flag (initValue 153 for every slice type) and
    sao_type_idx (200 / 185 / 160 for I / P / B: H.265 tables 9-5 / 9-6; entropy.cpp:196-208).
  * `sao_lambdas(tables, qp, cb_qp_offset)`: floor(256 * lambda2_tab[qp]) for luma and for chroma at the mapped Cb QP (sao.cpp:1229-1238)."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_PATH = os.path.join(ROOT, "tests", "golden", "host_tables.json")
SLICE_B, SLICE_P, SLICE_I = 0, 1, 2          # slice.h SliceType


def load(path=DEFAULT_PATH):
    import numpy as np
    d = json.load(open(path))
    return {"entropy_bits": np.asarray(d["entropy_bits"], dtype=np.uint32), "lambda2_tab": [float(v) for v in d["lambda2_tab"]],
            "chroma_scale": [int(v) for v in d["chroma_scale"]]}


def cabac_init_state(qp, init_value):
    qp = min(max(qp, 0), 51)
    slope = (init_value >> 4) * 5 - 45
    offset = ((init_value & 15) << 3) - 16
    init = min(max(1, ((slope * qp) >> 4) + offset), 126)
    mps = 1 if init >= 64 else 0
    return ((init - 64 if mps else 63 - init) << 1) + mps


def sao_contexts(slice_type, slice_qp):
    return cabac_init_state(slice_qp, 153), cabac_init_state(slice_qp, (160, 185, 200)[slice_type])


def sao_lambdas(tables, qp, cb_qp_offset=0, csp400=False, qp_min=0, qp_max=69):
    qc = qp + cb_qp_offset
    qc = min(max(qc, qp_min), qp_max) if csp400 else min(max(tables["chroma_scale"][min(max(qc, 0), 69)], qp_min), qp_max)
    return int(math.floor(256.0 * tables["lambda2_tab"][qp])), int(math.floor(256.0 * tables["lambda2_tab"][qc]))
