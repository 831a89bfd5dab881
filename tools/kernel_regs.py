"""Registers, spills, LDS and scratch of every kernel in libx265hip.so (measurement aid; runs without a GPU).

The library carries one clang offload bundle per translation unit (.hip_fatbin); this script pulls the gfx950 code objects out of them,
reads their metadata notes with llvm-readelf and prints one line per kernel: the occupancy a kernel can have is decided here
(512 vector registers per SIMD lane: waves per SIMD = 512 // vgprs, at most 8), before any profile is taken."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "x265-yuuki-asuna_amd", "libx265hip.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    blob = open(so, "rb").read()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in enumerate(code_objects(blob)):
            path = os.path.join(tmp, f"co{i}.elf")
            open(path, "wb").write(co)
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name).replace("x265hip::", "").replace("void ", "")
                rows.append((name, get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"), get("group_segment_fixed_size"),
                             get("private_segment_fixed_size"), get("max_flat_workgroup_size")))
    print(f"{'kernel':78s} {'vgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'lds':>6s} {'scratch':>7s} {'maxwg':>5s} {'waves/simd':>10s}")
    for r in sorted(set(rows)):
        if pat and not re.search(pat, r[0]):
            continue
        v = int(r[1]) if r[1].isdigit() else 0
        print(f"{r[0][:78]:78s} {r[1]:>5s} {r[2]:>6s} {r[3]:>6s} {r[4]:>6s} {r[5]:>7s} {r[6]:>5s} {min(8, 512 // max(v, 1)) if v else '?':>10}")


if __name__ == "__main__":
    main()
