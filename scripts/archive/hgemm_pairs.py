"""fp16 GEMM tile shapes, singles and pairs: python scripts/hgemm_pairs.py [M N K]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (4096, 1024, 1024)
names = {1: "128x128", 2: "64x64 split-K", 3: "256x128 (8 waves)", 4: "256x128 (4 waves, 128x64 each)"}
for mode, mname in ((4, "fwd, fp16 out"), (5, "fwd, fp16 + transposed out"), (1, "dgrad, 3 outputs")):
    for tile in (1, 3, 4, 11, 13, 14):
        us, err, ref = C.c_float(), C.c_float(), C.c_float()
        rc = lib.dqnhip_test_hgemm(mode, tile, M, N, K, 100, C.byref(us), C.byref(err), C.byref(ref))
        n = 2 if tile >= 10 else 1
        print("%-28s %-18s %s  rc %d  %7.2f us  %7.1f TF  err %.3g / %.3g" % (mname, names[tile % 10], "pair  " if n == 2 else "single", rc, us.value,
              n * 2.0 * M * N * K / (us.value * 1e-6) / 1e12 if us.value > 0 else 0, err.value, ref.value), flush=True)
