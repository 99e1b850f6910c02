"""Stand-alone timing of GEMM kernel variants of the test harness (tests/csrc: dqnhip_test_gemm), back to back on one stream.
usage: python scripts/gemm_variant_timing.py MODE ROWS N_OUT K_IN VARIANT [VARIANT ...]     (MODE: 0 fwd, 1 dgrad, 2 wgrad)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_gemm
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
mode, rows, n_out, k_in = (int(x) for x in sys.argv[1:5])
for rep in range(3):
    for v in (int(x) for x in sys.argv[5:]):
        us, err, ref = C.c_float(), C.c_float(), C.c_float()
        rc = fn(mode, v, rows, n_out, k_in, 1, 300, C.byref(us), C.byref(err), C.byref(ref))
        print("mode %d rows %d n_out %d k_in %d variant %d: rc %d  %.2f us per launch (back to back), max err %.2e of %.2e" % (mode, rows, n_out, k_in, v, rc, us.value, err.value, ref.value))
