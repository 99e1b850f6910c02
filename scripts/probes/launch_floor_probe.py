"""What does one link of a dependent kernel chain cost inside a replayed hipGraph, by launch shape?  (inside gpurun)"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from testlib import load_test
lib = load_test()
f = lib.dqnhip_test_launch_floor
f.restype = C.c_int; f.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_float)]
def run(variant, blocks, lds, chain=30, iters=200):
    us = C.c_float()
    rc = f(variant, blocks, lds, chain, iters, C.byref(us))
    return us.value if rc == 0 else float("nan")
print("us per kernel, chain of 30 in one graph (x 200 replays); columns: grid size")
grids = (16, 64, 256, 512, 1536)
print("%-58s" % "" + "".join("%9d" % g for g in grids))
rows = [("empty, 1 pointer arg", 0, 0), ("empty, 640-B kernarg", 1, 0), ("empty, 16 KB LDS", 2, 16384), ("empty, 48 KB LDS", 2, 49152),
        ("empty, 66 KB LDS", 2, 67584), ("empty, 640-B kernarg + 48 KB LDS", 3, 49152), ("one load->store round trip", 4, 0),
        ("round trip + 640-B kernarg + 48 KB LDS", 7, 49152), ("empty, 1024 threads", 8, 0), ("round trip, 1024 threads", 12, 0)]
for name, v, lds in rows:
    print("%-58s" % name + "".join("%9.2f" % run(v, g, lds) for g in grids), flush=True)
print("chain length (empty, 256 blocks): " + "  ".join("%d: %.2f" % (c, run(0, 256, 0, chain=c, iters=max(20, 3000 // c))) for c in (1, 2, 4, 8, 16, 30, 60, 120, 480)))
