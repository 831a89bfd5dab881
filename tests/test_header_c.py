"""include/x265hip.h as plain C: the header compiles with gcc -std=c99 and the header-only x265hip_surf_lookup finds every value of a
synthetic surface set in all four record formats (the layouts are written here from the header's prose, independently of the
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
    ctu_bytes = ((nc * ng + 63) // 64) * 64 * 720 if fmt == 3 else nc * ng * gb
    buf = np.zeros(nctu * ctu_bytes, np.uint8)
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
                            elif fmt == 2:    # chunk c of group g at row + (c * groups + g) * 16
                                a = row * 720 + ((o >> 4) * ng + g) * 16 + (o & 15)
                            else:             # blocks of 64 records (raster order m * groups + g), chunk c of slot l at block + (c * 64 + l) * 16
                                r = m * ng + g
                                a = ctu * ctu_bytes + (r >> 6) * 46080 + ((o >> 4) * 64 + (r & 63)) * 16 + (o & 15)
                            buf[a:a + size] = np.frombuffer((np.uint16(v) if size == 2 else np.int32(v)).tobytes(), np.uint8)
    return buf


@pytest.mark.parametrize("fmt", [0, 1, 2, 3])
def test_surf_lookup_in_plain_c(fmt, tmp_path):
    src, exe, data = tmp_path / "t.c", tmp_path / "t", tmp_path / "surf.bin"
    src.write_text(C_SRC)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rng, nctu = 5, 2
    _write(fmt, rng, nctu).tofile(data)
    out = subprocess.check_output([str(exe), str(data), str(fmt), str(rng), str(nctu)]).decode().split()
    assert int(out[0]) == nctu * 85 * (2 * rng + 1) ** 2 and int(out[1]) == 0


C_HOST = r"""
/* A C host of libx265hip.so without Python in the call path: links the library, checks the version string, drives two host-side
 * entries on data read from a file and prints their results (hex floats: bit-exact text), and confirms that a device entry fails
 * loudly - X265HIP_ENODEV with a message - when no gfx950 device is there. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "x265hip.h"
int main(int argc, char** argv)
{
    const char* ver = x265hip_version();
    if (!ver || !strstr(ver, "x265hip")) return 2;
    FILE* f = fopen(argv[1], "rb");
    int hdr[4];                                   /* width, height, part, number of partitions */
    if (fread(hdr, sizeof(int), 4, f) != 4) return 3;
    double span;
    if (fread(&span, sizeof(double), 1, f) != 1) return 3;
    const int n = hdr[3];
    uint64_t* sums = malloc(sizeof(uint64_t) * 8 * n);
    if (fread(sums, sizeof(uint64_t), 8 * (size_t)n, f) != 8 * (size_t)n) return 3;
    double* act = malloc(sizeof(double) * n); double* qp = malloc(sizeof(double) * n); int32_t* inv = malloc(sizeof(int32_t) * n);
    double avg = 0;
    x265hip_aq_hevc_offsets_params p;
    memset(&p, 0, sizeof(p));
    p.width = hdr[0]; p.height = hdr[1]; p.part = hdr[2]; p.qp_adaptation_range = span;
    p.sums = sums; p.activity = act; p.qp_offset = qp; p.avg_activity = &avg; p.inv_qscale = inv;
    if (x265hip_aq_hevc_offsets(&p) != 0) { fprintf(stderr, "%s\n", x265hip_last_error()); return 4; }
    printf("%a\n", avg);
    for (int i = 0; i < n; i++) printf("%a %a %d\n", act[i], qp[i], inv[i]);
    p.qp_adaptation_range = 0.5;                  /* out of [1, 6]: refused with a message */
    if (x265hip_aq_hevc_offsets(&p) != X265HIP_EINVAL || !strstr(x265hip_last_error(), "qp_adaptation_range")) return 5;
    if (argc > 2 && !strcmp(argv[2], "nodevice"))
    {
        uint64_t best[85];
        if (x265hip_me_best_reset(best, 85, NULL) != X265HIP_ENODEV || !x265hip_last_error()[0]) return 6;
    }
    return 0;
}
"""


def test_a_c_host_links_the_library_and_calls_it(tmp_path):
    """The drop-in boundary is a C ABI: a C99 program linked against libx265hip.so (no Python, no torch in the process) gets the same
    --hevc-aq numbers as the reference-pinned oracle and the documented error behaviour."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O
    F = importlib.import_module("x265-yuuki-asuna_amd.frames")
    A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
    if not os.path.exists(A.LIB_PATH):
        pytest.skip("libx265hip.so not built")
    width, height, part, span, depth = 250, 138, 16, 3.3333333333333335, 8
    y = F.synth_clip(width, height, 1, depth=depth, seed=9)[0][0]
    yp, stride, org, _, _ = F.pad_plane(y)
    sums = O.aq_hevc_quadrants(depth, yp, stride, org, width, height, part)
    n = sums.shape[0]
    blob = tmp_path / "sums.bin"
    with open(blob, "wb") as f:
        f.write(np.asarray([width, height, part, n], np.int32).tobytes())
        f.write(np.asarray([span], np.float64).tobytes())
        f.write(sums.tobytes())
    src = tmp_path / "host.c"
    src.write_text(C_HOST)
    exe = tmp_path / "host"
    libdir = os.path.dirname(A.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lx265hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    nodev = [] if torch.cuda.is_available() else ["nodevice"]
    out = subprocess.run([str(exe), str(blob)] + nodev, check=True, capture_output=True, text=True).stdout.split("\n")
    parts, act, qp, avg, inv, _, _ = O.aq_hevc_frame(depth, yp, stride, org, width, height, qg_size=16, qp_adaptation_range=span, weightp=False)
    at = int(parts[0] + parts[1])                                    # layer 2 = 16 x 16 partitions, the deepest for this quantisation group size
    assert float.fromhex(out[0]) == avg[2]
    for i in range(n):
        a, q, v = out[1 + i].split()
        assert float.fromhex(a) == act[at + i] and float.fromhex(q) == qp[at + i] and int(v) == inv[i], f"partition {i}"
