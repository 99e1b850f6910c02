"""forward tile variants on one 256x1024x1024 fp32 layer (single problem / two problems per launch)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
pkg = load_package(); lib = testlib.load_test()
fn = lib.dqnhip_test_gemm; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
def run(mode, variant, rows, n, k, groups):
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    rc = fn(mode, variant, rows, n, k, groups, 200, C.byref(us), C.byref(err), C.byref(ref))
    print("mode %d var %2d rows %4d groups %d: rc %d %7.2f us  err %.2e" % (mode, variant, rows, groups, rc, us.value, err.value), flush=True)
for rows in (256, 64):
    for v in (15, 20, 16):
        for g in (1, 2):
            run(0, v, rows, 1024, 1024, g)
for v in (1, 5, 2, 6):
    for g in (1,):
        run(1, v, 256, 1024, 1024, g)
