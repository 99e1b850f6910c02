import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_chain; fn.restype = C.c_int
fn.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_int32)]
for rep in range(3):
    for layers in (4, 8):
        for mp in (0, 4):
            a, b, d = C.c_float(), C.c_float(), C.c_float(); g = C.c_int32()
            rc = fn(layers, mp, 300, C.byref(a), C.byref(b), C.byref(d), C.byref(g))
            print("layers %d %s: %.2f us per layer (launches)  maxdiff %.3g rc %d" % (layers, "write-through stores" if mp else "plain stores        ", a.value / layers, d.value, rc), flush=True)
