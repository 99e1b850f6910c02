"""k_adam_soft on 1/N of a net's arena (what a rank of a sharded-optimiser group would run): python scripts/adam_slice_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_adam; fn.restype = C.c_int
fn.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
N = 3220544          # one BASELINE net's arena (critic, padded)
for touch in (0, 256):
    for div in (1, 2, 4, 8):
        n = N // div // 1024 * 1024
        blocks = min(1536, n // 4 // 256)
        us = C.c_float()
        rc = fn(n, 10, blocks, 200, touch, C.byref(us))
        print("arena / %d = %8d params, %4d blocks, %3d MB of other traffic between passes: %6.2f us  rc %d" % (div, n, blocks, touch, us.value, rc), flush=True)
