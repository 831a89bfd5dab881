"""Debug helper (GPU box): find which HIP table slots change the real encoder's bitstream."""
import ctypes, importlib, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
import test_ref_encoder as T
A = importlib.import_module("x265-yuuki-asuna_amd.hipabi")
F = T.F
spec = H.spec
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
preset = sys.argv[2] if len(sys.argv) > 2 else "ultrafast"
lib = T.ref_lib(depth, ROOT)
clip = F.synth_clip(128, 64, 3, depth=depth, seed=32)
base, _, _ = T.encode(lib, clip, 128, 64, preset, T.OPTS)
L = A.lib()
A.set_entropy_bits(list((ctypes.c_uint32 * 128).in_dll(lib, "x265_entropyStateBits")))
L.x265hip_setup_primitives.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
fields = sorted({H.field_of(p) for p in spec.SLOTS})
only = sys.argv[3].split(",") if len(sys.argv) > 3 else fields          # optional: the fields to try one by one
def run(selected):
    def fill(tab, nbytes, d):
        tmp = (ctypes.c_void_p * spec.TABLE_PTRS)()
        L.x265hip_setup_primitives(ctypes.byref(tmp), spec.TABLE_BYTES, d)
        dst = (ctypes.c_void_p * spec.TABLE_PTRS).from_address(tab)
        cnt = 0
        for path, (td, i) in spec.SLOTS.items():
            if tmp[i] and H.field_of(path) in selected:
                dst[i] = tmp[i]; cnt += 1
        return cnt
    cb = T.FILL(fill)
    got, _, filled = T.encode(lib, clip, 128, 64, preset, T.OPTS, ctypes.cast(cb, ctypes.c_void_p))
    return got == base, filled
ok, n = run(set(fields)); print("all fields:", ok, n)
bad = []
for f in only:
    ok, n = run({f})
    if n and not ok:
        bad.append(f); print("MISMATCH with only", f, "(", n, "slots )")
print("bad fields:", bad)
