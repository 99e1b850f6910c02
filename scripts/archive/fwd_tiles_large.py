"""forward tile variants of the fp32 gemm_fwd_lds family on large-row problems (env front-end at 1024 / 2048 workers, fp32 updates at
minibatch 4096): rows x 1024 x 1024, one problem per launch, 200 launches each"""
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
    tf = 2.0 * rows * n * k * groups / (us.value * 1e-6) / 1e12 if us.value > 0 else 0
    print("mode %d var %2d rows %4d groups %d: rc %d %7.2f us  %6.1f TF  err %.2e" % (mode, variant, rows, groups, rc, us.value, tf, err.value), flush=True)
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [13, 12, 14, 19, 17]
for rows in (1024, 2048, 4096):
    for v in variants:
        run(0, v, rows, 1024, 1024, 1)
