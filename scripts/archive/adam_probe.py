"""Fused clip+Adam+soft-update pass: variants x grid sizes x interleaved traffic (dqnhip_test_adam)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_adam; fn.restype = C.c_int
fn.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
N = 3220544          # one BASELINE net's arena (critic, padded)
print("variant blocks touch_mb  us   TB/s(36 B/param)")
for touch in (0, 64, 256):
    for variant in (10, 11, 20, 21, 40, 41):
        for blocks in (512, 1024, 2048, 4096, 8192):
            us = C.c_float()
            rc = fn(N, variant, blocks, 50, touch, C.byref(us))
            print("%5d %7d %6d %7.2f %6.2f" % (variant, blocks, touch, us.value, N * 36 / us.value / 1e6) if rc == 0 else ("rc", rc), flush=True)
