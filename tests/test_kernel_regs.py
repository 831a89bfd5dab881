"""Register budget of the kernels the default step launches, read from the built library without a GPU (tools/kernel_regs.py).

Round 3 found the 32x32 TU kernels at 284 - 401 vector registers - ONE wavefront per SIMD, 103 us instead of 73 us per 4K picture - only by
reading the code object's notes; nothing in the functional tests can see that.  This pins what was measured: the occupancy each hot
kernel was tuned for, and no scratch memory on the serial path of the SAO decision."""
import importlib.util
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "x265-yuuki-asuna_amd", "libx265hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernels():
    if not (os.path.exists(SO) and os.path.exists(READELF)):
        pytest.skip("library or llvm-readelf not present")
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "tools", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    import tempfile
    out = {}
    blob = open(SO, "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in enumerate(kr.code_objects(blob)):
            path = os.path.join(tmp, f"co{i}.elf")
            open(path, "wb").write(co)
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]
                name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name).replace("x265hip::", "").replace("void ", "")
                out[name] = {"vgpr": int(get("vgpr_count")), "vspill": int(get("vgpr_spill_count")), "scratch": int(get("private_segment_fixed_size"))}
    assert len(out) > 300, "the code objects of the library could not be read"
    return out


def test_hot_kernels_keep_the_occupancy_they_were_tuned_for():
    k = _kernels()
    waves = lambda name: min(8, 512 // k[name]["vgpr"])
    # the uni-predictive 16 / 32 point TU stages of the default step: two wavefronts per SIMD, nothing spilled (profiles/r03_tail_kernels.txt)
    for name in ("inter_recon_kernel<unsigned char, 32, false, false>", "inter_recon_kernel<unsigned char, 16, true, false>"):
        assert waves(name) >= 2 and k[name]["vspill"] == 0 and k[name]["scratch"] == 0, (name, k[name])
    # the exhaustive search the bench times: two workgroups of four wavefronts per CU by design (194 registers), no spills
    for me in ("me_ctu_c_kernel<true, true, 3, 3>", "me_ctu_c_kernel<true, true, 2, 3>"):          # block-major (the default) / chunk-major records
        assert waves(me) == 2 and k[me]["vspill"] == 0 and k[me]["scratch"] == 0, (me, k[me])
    # the serial pass of the SAO decision: one wavefront's chain IS the step - no scratch, no spills
    for name in ("sao_rdo_rows2_kernel<3>", "sao_rdo_rows2_kernel<1>"):
        assert k[name]["vspill"] == 0 and k[name]["scratch"] == 0 and k[name]["vgpr"] <= 128, (name, k[name])
    # the streaming kernels of the step run at full occupancy
    for name in ("sao_stats_kernel<unsigned char>", "sao_apply_kernel<unsigned char>", "lowres_intra_kernel<unsigned char>", "phase_luma_kernel<unsigned char>",
                 "extend_border_kernel<unsigned char>", "deblock_luma_kernel<unsigned char, 0>"):
        assert waves(name) >= 4 and k[name]["vspill"] == 0, (name, k[name])
