"""Run one GEMM variant a few times (for rocprofv3 PMC passes): mode variant groups [iters]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_gemm; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
mode, variant, groups = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
us, err, ref = C.c_float(), C.c_float(), C.c_float()
rc = fn(mode, variant, 256, 1024, 1024, groups, iters, C.byref(us), C.byref(err), C.byref(ref))
print(rc, us.value, err.value)
