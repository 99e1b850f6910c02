"""Every MFMA GEMM kernel variant against a naive one-thread-per-output fp32 reference on
uniform [-1,1) data (asymmetric operands: a transposed tile or a swapped fragment shows up as
an O(1) error).  Tolerance: fp32 summation-order noise only."""
import ctypes as C

import pytest

import testlib

pytestmark = pytest.mark.gpu

FWD, DGRAD, WGRAD = 0, 1, 2


def _run(pkg, mode, variant, rows, n_out, k_in, groups=1):
    lib = testlib.load_test()
    fn = lib.dqnhip_test_gemm
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32] * 7 + [C.POINTER(C.c_float)] * 3
    us, err, ref = C.c_float(), C.c_float(), C.c_float()
    rc = fn(mode, variant, rows, n_out, k_in, groups, 2, C.byref(us), C.byref(err), C.byref(ref))
    assert rc == 0
    return err.value, ref.value


# (rows, n_out, k_in): BASELINE layer, reference-default layers, padded first layers, B=32
SHAPES = [(256, 1024, 1024), (32, 512, 1024), (32, 128, 256), (64, 1024, 64), (256, 1024, 128),
          (32, 64, 64), (96, 192, 320)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode,variant", [(FWD, 1), (FWD, 2), (FWD, 10), (FWD, 12), (FWD, 15), (FWD, 16), (FWD, 17), (FWD, 20), (DGRAD, 1), (DGRAD, 2),
                                          (DGRAD, 5), (DGRAD, 6), (DGRAD, 7), (DGRAD, 8), (WGRAD, 1), (WGRAD, 2), (WGRAD, 3), (FWD, 0), (DGRAD, 0), (WGRAD, 0)])
def test_gemm_variant_vs_naive(pkg, gpu, mode, variant, shape):
    rows, n_out, k_in = shape
    # tile divisibility of each variant (the learner only launches shapes that satisfy them)
    if mode == FWD and variant == 2 and n_out % 64:
        pytest.skip("64-wide P tile")
    if variant == 0 and (n_out % 64 or k_in % 64):
        pytest.skip("LDS-staged family needs 64-multiples")
    if mode == DGRAD and variant in (2, 6) and rows % 32:
        pytest.skip("32-row Q tile")
    if mode == FWD and variant in (10, 12, 15, 16, 17, 20) and (k_in % 256 or k_in < 512 or (variant in (12, 17) and n_out % 64)):
        pytest.skip("coalesced forward needs K % 256 == 0, K >= 512")
    if mode == DGRAD and variant in (5, 6) and (n_out % 256 or n_out < 512 or k_in % 64):
        pytest.skip("coalesced dgrad needs N % 256 == 0, N >= 512")
    if mode == DGRAD and variant == 8 and (k_in % 32 or n_out % 64):
        pytest.skip("32-column tiles, reduction split over four waves in 16-deep blocks")
    if mode == WGRAD and variant == 3 and (n_out % 64 or k_in % 64):
        pytest.skip("64 x 64 tiles")
    for groups in (1, 2):
        err, ref = _run(pkg, mode, variant, rows, n_out, k_in, groups)
        red = {FWD: k_in, DGRAD: n_out, WGRAD: rows}[mode]
        assert ref > 1.0
        assert err <= 4e-7 * red ** 0.5 * ref + 1e-6, (mode, variant, shape, groups, err, ref)
