"""fp32 backward at 256 rows: is 'dgrad chain, then all wgrads of the net in one launch' cheaper than one gemm_bwd_seq per layer?
Times the stand-alone pieces (200 launches each, warm): dgrad_lds<1,1>, wgrad_direct<1,1> with 1 / 2 / 3 / 4 problems per launch."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
pkg = load_package(); lib = testlib.load_test()
fn = lib.dqnhip_test_gemm; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
def run(mode, variant, rows, n, k, groups, tag):
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    rc = fn(mode, variant, rows, n, k, groups, 200, C.byref(us), C.byref(err), C.byref(ref))
    print("%-28s rows %4d groups %d: rc %d %7.2f us  err %.2e" % (tag, rows, groups, rc, us.value, err.value), flush=True)
for rows in (256, 512):
    run(1, 5, rows, 1024, 1024, 1, "dgrad_lds<1,1>")
    run(1, 5, rows, 1024, 1024, 2, "dgrad_lds<1,1>")
    for g in (1, 2, 3, 4):
        run(2, 1, rows, 1024, 1024, g, "wgrad_direct<1,1>")
    run(2, 2, rows, 1024, 128, 1, "wgrad_narrow<1>")
