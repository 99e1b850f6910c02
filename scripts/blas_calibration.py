"""Calibration, not product: the vendor BLAS (hipBLASLt / rocBLAS behind torch.matmul) on the tower's layer
shapes next to this library's kernels, same box, same call, back-to-back launches on one stream.

  python scripts/blas_calibration.py [iters]

fp32 rows:  256 x 1024 x 1024 forward (Y = X W^T), dgrad (dX = dY W), wgrad (dW = dY^T X)  -- exact fp32 in both
fp16 rows:  4096 / 512 x 1024 x 1024, fp16 operands, fp32 accumulate, fp16 result
torch carries the vendor library here only; nothing in the library under test calls it.
"""
import ctypes as C, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import testlib
lib = testlib.load_test()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False


def time_torch(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def ours_gemm(mode, variant, rows, n, k, groups=1):
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    fn = lib.dqnhip_test_gemm
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
    rc = fn(mode, variant, rows, n, k, groups, iters, C.byref(us), C.byref(err), C.byref(ref))
    assert rc == 0, rc
    return us.value


def ours_hgemm(mode, tile, M, N, K):
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    rc = lib.dqnhip_test_hgemm(mode, tile, M, N, K, iters, C.byref(us), C.byref(err), C.byref(ref))
    assert rc == 0, rc
    return us.value


out = []

def row(name, flop, t_blas, t_ours, peak):
    r = {"case": name, "blas_us": round(t_blas, 2), "ours_us": round(t_ours, 2),
         "blas_TF": round(flop / t_blas / 1e6, 1), "ours_TF": round(flop / t_ours / 1e6, 1),
         "ours_frac_of_peak": round(flop / t_ours / 1e6 / peak, 3)}
    out.append(r)
    print(json.dumps(r), flush=True)


# ---- fp32, 256-row minibatch ------------------------------------------------------------------------------------
B, N, K = 256, 1024, 1024
X = torch.randn(B, K, device=dev); W = torch.randn(N, K, device=dev); dY = torch.randn(B, N, device=dev)
Y = torch.empty(B, N, device=dev); dX = torch.empty(B, K, device=dev); dW = torch.empty(N, K, device=dev)
fl = 2.0 * B * N * K
row("fp32 fwd 256x1024x1024", fl, time_torch(lambda: torch.matmul(X, W.t(), out=Y), iters), ours_gemm(0, 11, B, N, K), 157.3)
row("fp32 dgrad 256x1024x1024", fl, time_torch(lambda: torch.matmul(dY, W, out=dX), iters), ours_gemm(1, 5, B, N, K), 157.3)
row("fp32 wgrad 1024x1024 k=256", fl, time_torch(lambda: torch.matmul(dY.t(), X, out=dW), iters), ours_gemm(2, 1, B, N, K), 157.3)

# ---- fp16 operands, fp32 accumulate -------------------------------------------------------------------------------
for M in (4096, 512):
    A = torch.randn(M, K, device=dev, dtype=torch.float16); Wh = torch.randn(N, K, device=dev, dtype=torch.float16)
    Ch = torch.empty(M, N, device=dev, dtype=torch.float16)
    fl = 2.0 * M * N * K
    row("fp16 fwd %dx1024x1024" % M, fl, time_torch(lambda: torch.matmul(A, Wh.t(), out=Ch), iters), ours_hgemm(4, 0, M, N, K), 2500.0)
    dYt = torch.randn(N, M, device=dev, dtype=torch.float16); Xt = torch.randn(K, M, device=dev, dtype=torch.float16)
    dWh = torch.empty(N, K, device=dev, dtype=torch.float16)
    # wgrad-shaped: 1024 x 1024 outputs, reduction over the M minibatch rows (both operands reduction-contiguous, as ours)
    row("fp16 wgrad 1024x1024 k=%d" % M, fl, time_torch(lambda: torch.matmul(dYt, Xt.t(), out=dWh), iters), ours_hgemm(2, 0, N, K, M), 2500.0)

with open(os.path.join(ROOT, "gpurun_out", "blas_calibration.json"), "w") as f:
    json.dump(out, f, indent=1)
