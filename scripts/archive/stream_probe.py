import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_gemm; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
ns = os.environ.get("DQNHIP_BENCH_STREAMS", "1")
for mode, variant, name in ((0, 10, "FWD 32x32 lds"), (0, 12, "FWD 64x32 lds"), (1, 1, "DGRAD 64x16"), (2, 1, "WGRAD 64x64")):
    for groups in (1, 2):
        us = C.c_float(); e = C.c_float(); r = C.c_float()
        rc = fn(mode, variant, 256, 1024, 1024, groups, 50, C.byref(us), C.byref(e), C.byref(r))
        n = int(ns) * groups
        print("streams %s %-14s g%d: %7.2f us per round of %d GEMMs -> %5.2f us/GEMM" % (ns, name, groups, us.value, n, us.value / n), flush=True)
