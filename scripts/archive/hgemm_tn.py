"""wgrad-shaped fp16 GEMM: k-major operands (mode 2, needs the transposed panels) vs reduction-major operands (mode 6,
transposing LDS reads): python scripts/hgemm_tn.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
for (M, N, K) in ((64, 64, 128), (128, 128, 128), (128, 128, 64), (256, 128, 512), (1024, 1024, 512), (1024, 1024, 4096), (1024, 128, 4096)):
    for mode in (2, 6):
        for tile in (1, 2):
            if tile == 2 and K % 128: continue
            us, err, ref = C.c_float(), C.c_float(), C.c_float()
            rc = lib.dqnhip_test_hgemm(mode, tile, M, N, K, 50, C.byref(us), C.byref(err), C.byref(ref))
            print("M %4d N %4d K %4d  %s  tile %s  rc %d  %7.2f us  err %.3g / %.3g" % (M, N, K, "k-major        " if mode == 2 else "reduction-major", "128x128" if tile == 1 else "64x64  ", rc, us.value, err.value, ref.value), flush=True)
