"""GEMM kernel A/B on the GPU: python scripts/gemm_tune.py [rows n_out k_in]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_gemm
fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
rows, n_out, k_in = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 1024, 1024)
names = {0: "FWD", 1: "DGRAD", 2: "WGRAD"}
variants = {0: [], 1: [1, 5, 2, 6], 2: [1]}
flops = 2.0 * rows * n_out * k_in
for rep in range(1):
    for mode in (0, 1, 2):
        for v in variants[mode]:
            for groups in (1, 2, 4):
                us, err, ref = C.c_float(), C.c_float(), C.c_float()
                rc = fn(mode, v, rows, n_out, k_in, groups, 50, C.byref(us), C.byref(err), C.byref(ref))
                tf = flops * groups / (us.value * 1e-6) / 1e12 if rc == 0 and us.value > 0 else 0
                print("%-5s v%d g%d  rc=%d  %8.2f us/launch  %6.1f TF  err %.2e (ref max %.1f)" % (
                    names[mode], v, groups, rc, us.value, tf, err.value, ref.value), flush=True)
