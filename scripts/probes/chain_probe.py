"""Persistent forward chain vs separate launches (dqnhip_test_chain)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(ROOT, "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_chain; fn.restype = C.c_int
fn.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int32)]
print("layers map  us(launches)  us(persistent)  per-layer: launches / persistent   max|diff|  gave_up")
for rep in range(2):
    for layers in (1, 2, 4, 8):
        # bit 0: slab per XCD, bit 1: sc1 write-through hand-off (no release fence), bit 3 (round 5): weights requested BEFORE the
        # poll + activations through sc1 loads instead of the consumer's acquire (implies bit 1)
        for mp in (0, 2, 3, 10, 11):
            a, b, d = C.c_float(), C.c_float(), C.c_float(); g = C.c_int32()
            rc = fn(layers, mp, 200, C.byref(a), C.byref(b), C.byref(d), C.byref(g))
            print("%4d %4d %12.2f %14.2f %14.2f / %.2f %14.3g %6d  rc=%d" % (layers, mp, a.value, b.value, a.value / layers, b.value / layers, d.value, g.value, rc), flush=True)
