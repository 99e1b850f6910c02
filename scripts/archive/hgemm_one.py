"""one fp16 GEMM shape, many launches (for rocprofv3 --pmc): hgemm_one.py mode tile M N K iters"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
mode, tile, M, N, K, iters = [int(x) for x in sys.argv[1:7]]
us, err, ref = C.c_float(), C.c_float(), C.c_float()
rc = lib.dqnhip_test_hgemm(mode, tile, M, N, K, iters, C.byref(us), C.byref(err), C.byref(ref))
print("rc", rc, "us", us.value, "TF", 2.0 * M * N * K / (us.value * 1e-6) / 1e12, "err", err.value)
