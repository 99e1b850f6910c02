"""layer backward without transposed panels: python scripts/hgemm_backward.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_hgemm_backward; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_float)] * 3
for (rows, n_out, k_in) in ((128, 128, 128), (256, 256, 128), (256, 1024, 1024), (512, 1024, 1024), (4096, 1024, 1024), (4096, 1024, 128)):
    us = (C.c_float * 3)(); err, ref = C.c_float(), C.c_float()
    rc = fn(rows, n_out, k_in, 50, us, C.byref(err), C.byref(ref))
    print("rows %4d n_out %4d k_in %4d  rc %d  dgrad %7.2f us  wgrad %7.2f us  pair %7.2f us  err %.3g / %.3g" % (rows, n_out, k_in, rc, us[0], us[1], us[2], err.value, ref.value), flush=True)
