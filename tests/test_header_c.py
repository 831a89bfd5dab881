"""include/x265hip.h as plain C: the header compiles with gcc -std=c99 and the header-only x265hip_surf_lookup finds every value of a
synthetic surface set in all three record formats (the layouts are written here from the header's prose, independently of the
function)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r"""
#include <stdio.h>
#include <stdlib.h>
#include "x265hip.h"
int main(int argc, char** argv)
{
    int fmt = atoi(argv[2]), range = atoi(argv[3]), nctu = atoi(argv[4]);
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* buf = malloc(n);
    if (fread(buf, 1, n, f) != (size_t)n) return 2;
    static const int npu[4] = { 64, 16, 4, 1 };
    long bad = 0, checked = 0;
    for (int ctu = 0; ctu < nctu; ctu++)
        for (int level = 0; level < 4; level++)
            for (int z = 0; z < npu[level]; z++)
                for (int dy = -range; dy <= range; dy++)
                    for (int dx = -range; dx <= range; dx++)
                    {
                        int want = (ctu * 131 + level * 17 + z * 7 + (dy + range) * 3 + (dx + range) * 5) % (level < 2 ? 65521 : 1000003);
                        bad += x265hip_surf_lookup(buf, fmt, range, ctu, level, z, dx, dy) != want;
                        checked++;
                    }
    printf("%ld %ld\n", checked, bad);
    return bad != 0;
}
"""


def _write(fmt, rng, nctu):
    nc = 2 * rng + 1
    ng = (nc + 3) // 4
    base_pu = (0, 64, 80, 84)
    npu = (64, 16, 4, 1)
    pbase = (0, 512, 640, 704)
    gb = 1360 if fmt == 0 else 720
    buf = np.zeros(nctu * nc * ng * gb, np.uint8)
    for ctu in range(nctu):
        for level in range(4):
            for z in range(npu[level]):
                for m in range(nc):
                    for c in range(nc):
                        v = (ctu * 131 + level * 17 + z * 7 + m * 3 + c * 5) % (65521 if level < 2 else 1000003)
                        g, k = c >> 2, c & 3
                        row = (ctu * nc + m) * ng
                        if fmt == 0:          # int32 [ctu][mvy][group][85][4]
                            o = (row + g) * 1360 + ((base_pu[level] + z) * 4 + k) * 4
                            buf[o:o + 4] = np.frombuffer(np.int32(v).tobytes(), np.uint8)
                        else:
                            size = 2 if level < 2 else 4
                            o = pbase[level] + (z * 4 + k) * size          # byte inside the 720-byte record
                            if fmt == 1:      # record-contiguous
                                a = (row + g) * 720 + o
                            else:             # chunk c of group g at row + (c * groups + g) * 16
                                a = row * 720 + ((o >> 4) * ng + g) * 16 + (o & 15)
                            buf[a:a + size] = np.frombuffer((np.uint16(v) if size == 2 else np.int32(v)).tobytes(), np.uint8)
    return buf


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_surf_lookup_in_plain_c(fmt, tmp_path):
    src, exe, data = tmp_path / "t.c", tmp_path / "t", tmp_path / "surf.bin"
    src.write_text(C_SRC)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rng, nctu = 5, 2
    _write(fmt, rng, nctu).tofile(data)
    out = subprocess.check_output([str(exe), str(data), str(fmt), str(rng), str(nctu)]).decode().split()
    assert int(out[0]) == nctu * 85 * (2 * rng + 1) ** 2 and int(out[1]) == 0
