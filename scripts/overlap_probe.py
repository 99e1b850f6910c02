"""Does the optimiser pass hide under MFMA-bound launches?  (csrc/gemm_bench.hip dqnhip_test_overlap)
   python scripts/overlap_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_overlap; fn.restype = C.c_int
fn.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
names = ["L launches alone", "optimiser pass alone", "serial (today)", "riders, 48-KiB LDS (co-resident)", "two streams", "riders, 96-KiB LDS (control)"]
for layers, params, rb in ((3, 3 * 1048576, 256), (3, 3 * 1048576, 512), (3, 3 * 1048576, 768), (4, 4 * 1048576, 512), (3, 3 * 262144, 256)):
    us = (C.c_float * 6)()
    for rep in range(2):
        rc = fn(layers, params, rb, 200, us)
        print("layers %d, optimiser params %d (%.0f MB of traffic), %d rider blocks per launch  rc %d" % (layers, params, params * 36 / 1e6, rb, rc))
        for n, u in zip(names, us):
            print("    %-36s %8.2f us" % (n, u), flush=True)
