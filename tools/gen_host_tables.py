#!/usr/bin/env python3
"""tests/golden/host_tables.json: HOST-built tables an encoder hands to the device stages that price bits - read from the reference build
(oracle/_ref/libx265ref8.so, i.e. only where /root/reference exists; the fixture travels):
  entropy_bits  g_entropyBits[128] (encoder/entropy.cpp:2611): per-state CABAC bit costs, the `entropy_bits` input of x265hip_sao_rdo
  lambda2_tab   x265_lambda2_tab[QP_MAX_MAX + 1] (common/constants.cpp): SAO's lambda = floor(256 * lambda2_tab[qp]) (sao.cpp:1237-1238)
  entropy_state_bits  x265_entropyStateBits[128] (encoder/entropy.cpp:2647): the same costs with the CABAC transition packed into the top byte -
                the form the oracle's costCoeffNxN restatement reads (oracle/x265_oracle_host.c); x265hip_set_entropy_bits takes either
  chroma_scale  g_chromaScale[70] (common/constants.cpp): the 4:2:0 chroma QP mapping used for the chroma lambda (sao.cpp:1232-1236)
A fixture is DATA (inputs the host owns); no reference source text is stored."""
import ctypes
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libx265ref8.so"))
    lib.x265ref_entropy_bits_table.restype = ctypes.POINTER(ctypes.c_uint32)
    bits = [int(lib.x265ref_entropy_bits_table()[i]) for i in range(128)]
    state_bits = [int(v) for v in (ctypes.c_uint32 * 128).in_dll(lib, "x265_entropyStateBits")]
    assert all((a & 0xFFFFFF) == b for a, b in zip(state_bits, bits))
    lam2 = (ctypes.c_double * 70).in_dll(lib, "_ZN4x26516x265_lambda2_tabE")
    cscale = (ctypes.c_uint8 * 70).in_dll(lib, "_ZN4x26513g_chromaScaleE")
    out = {"source": "oracle/_ref/libx265ref8.so (x265 3.5 compiled from /root/reference): g_entropyBits, x265_entropyStateBits, x265_lambda2_tab, g_chromaScale; tools/gen_host_tables.py",
           "entropy_bits": bits, "entropy_state_bits": state_bits, "lambda2_tab": [float(v) for v in lam2], "chroma_scale": [int(v) for v in cscale]}
    path = os.path.join(ROOT, "tests", "golden", "host_tables.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
