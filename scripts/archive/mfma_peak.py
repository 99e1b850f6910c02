"""MFMA fp32 issue-rate probe on the GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_gemm; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
for nacc in (1, 2, 4):
    for blocks in (256, 512, 1024):
        for loops in (64, 1024):
            us = C.c_float()
            fn(99, nacc, blocks, loops, 0, 1, 20, C.byref(us), None, None)
            nm = blocks * 4 * loops * 4 * nacc          # MFMAs
            tf = nm * 2048 / (us.value * 1e-6) / 1e12
            cyc = us.value * 1e-6 * 2.4e9 / (nm / 1024)  # cycles per MFMA per SIMD at 2.4 GHz if all 1024 SIMDs busy
            print("nacc %d blocks %4d loops %4d: %8.2f us  %6.1f TF  (%.1f cyc@2.4GHz per MFMA per SIMD)" % (nacc, blocks, loops, us.value, tf, cyc), flush=True)
print("v_mfma_f32_32x32x16_f16:")
for nacc in (1, 2, 4):
    for blocks in (256, 512):
        for loops in (256, 4096):
            us = C.c_float()
            fn(99, 10 + nacc, blocks, loops, 0, 1, 20, C.byref(us), None, None)
            nm = blocks * 4 * loops * 4 * nacc
            tf = nm * 32768 / (us.value * 1e-6) / 1e12
            cyc = us.value * 1e-6 * 2.4e9 / (nm / 1024)
            print("nacc %d blocks %4d loops %4d: %8.2f us  %7.1f TF  (%.1f cyc@2.4GHz per MFMA per SIMD)" % (nacc, blocks, loops, us.value, tf, cyc), flush=True)
