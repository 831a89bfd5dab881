"""The reference pin must be REBUILDABLE: `make -C oracle ref8 OUT=<tmp>` from a clean directory has to succeed and give the
very libraries `oracle/_ref` ships (round-2 verdict, weak 1: a shim rule had lost its compile line and only an incremental build
still worked).  Needs the reference sources, i.e. this container; skipped on the GPU box."""
import filecmp
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/source"
SHIPPED = os.path.join(ROOT, "oracle", "_ref")

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources are only present in the build container")]


def test_makefile_has_a_compile_command_for_every_shim():
    """Static guard, cheap: every ref_*.cpp of oracle/ is linked into a library and is covered by a rule with a compile line."""
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    shims = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "oracle")) if f.startswith("ref_") and f.endswith(".cpp"))
    assert shims, "no shims found"
    for s in shims:
        assert f"obj$(3)/{s}.o" in mk, f"{s}.o is linked into no library"          # $(3) = the build tag (8, 10, 12, 8v3, 10v3)
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-n", "-B", "ref8", "OUT=/tmp/_x265hip_recipe_dry"],
                         capture_output=True, text=True, check=True).stdout
    for s in shims:
        assert any(f"{s}.cpp" in ln and " -c " in ln for ln in out.splitlines()), f"no compile command is issued for {s}.cpp"


def test_ref8_builds_from_clean_and_equals_the_shipped_libraries(tmp_path):
    if not os.path.exists(os.path.join(SHIPPED, "libx265ref8.so")):
        pytest.skip("oracle/_ref not built yet (python -c 'import __graft_entry__ as g; g.build()')")
    out = str(tmp_path / "ref")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), f"-j{min(16, os.cpu_count() or 4)}", "ref8", f"OUT={out}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, "clean `make ref8` failed:\n" + r.stdout[-2000:] + r.stderr[-4000:]
    try:
        for lib in ("libx265ref8.so", "libx265ref8_seam.so"):
            assert os.path.exists(os.path.join(out, lib)), lib + " not produced"
            # g++ is deterministic for identical sources / flags / paths of the INPUTS; the object directory does not enter the code
            assert filecmp.cmp(os.path.join(out, lib), os.path.join(SHIPPED, lib), shallow=False), \
                lib + ": a clean build differs from the shipped pin - rebuild oracle/_ref (make -C oracle ref) and rerun the pins"
    finally:
        shutil.rmtree(out, ignore_errors=True)
