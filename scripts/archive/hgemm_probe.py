"""fp16 GEMM family: correctness + timing probe (dqnhip_test_hgemm)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
lib.dqnhip_test_hgemm.restype = C.c_int
def run(mode, tile, M, N, K, iters=20):
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    rc = lib.dqnhip_test_hgemm(mode, tile, M, N, K, iters, C.byref(us), C.byref(err), C.byref(ref))
    tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12 if us.value > 0 else 0
    print(f"mode {mode} tile {tile} M{M} N{N} K{K}: rc={rc} {us.value:8.2f} us {tf:7.1f} TF  err {err.value:.3e} ref {ref.value:.3e}", flush=True)
cases = [(3, 0, 256, 192, 64), (0, 1, 128, 128, 64), (0, 2, 64, 64, 64), (0, 1, 256, 256, 192), (0, 2, 128, 192, 256), (1, 1, 256, 128, 128), (1, 2, 128, 128, 128), (2, 2, 128, 128, 512), (2, 1, 128, 256, 512)]
cases += [(4, 1, 4096, 1024, 1024), (5, 1, 4096, 1024, 1024), (4, 1, 4096, 1024, 2048), (4, 1, 4096, 1024, 4096), (0, 1, 4096, 1024, 1024), (1, 1, 4096, 1024, 1024), (2, 1, 1024, 1024, 4096), (2, 2, 1024, 1024, 4096), (0, 2, 512, 1024, 1024), (0, 2, 256, 1024, 1024), (0, 1, 4096, 1024, 128), (2, 2, 1024, 128, 4096), (0, 1, 8192, 8192, 8192)]
for c in cases:
    run(*c)
