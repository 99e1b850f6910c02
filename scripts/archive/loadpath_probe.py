"""CU load-path probe (libdqnhip_test.so): bytes per second the CUs pull from L2 / L1 in the fp16 GEMM's stage shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
fn = lib.dqnhip_test_loadpath; fn.restype = C.c_int
fn.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_float)] * 2
names = {0: "register loads", 1: "LDS-DMA", 2: "LDS-DMA + fragment reads", 3: "fragment reads alone",
         4: "reg-staged ds_write + reads", 5: "reg-staged ds_write",
         6: "half DMA + half direct + reads", 7: "half DMA + half direct",
         8: "loader wave DMA + 4 reader waves", 9: "GEMM-pattern fragment reads alone"}
for blocks in (256, 512):
    for region_kb in (1024,):
        for mode in (0, 1, 2, 3, 9):
            if blocks == 512 and mode != 0:
                continue                                  # 128 KiB of LDS: one workgroup per CU
            us, tb = C.c_float(), C.c_float()
            rc = fn(mode, blocks, region_kb, 64, 50, C.byref(us), C.byref(tb))
            print("blocks %3d region/XCD %5d KiB  %-26s rc %d  %7.2f us  %6.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz"
                  % (blocks, region_kb, names[mode], rc, us.value, tb.value, tb.value * 1e12 / 256 / 2.4e9), flush=True)
