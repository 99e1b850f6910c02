/*
 * dqnhip_internal.h — test / tuning hooks exported by libdqnhip.so.  Not part of the
 * drop-in boundary (include/dqnhip.h); used by tests/ and scripts/gemm_tune.py only.
 */
#ifndef DQNHIP_INTERNAL_H_
#define DQNHIP_INTERNAL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Run one GEMM kernel variant of the tower layers on random data (uniform [-1,1)),
 * check it against a naive one-thread-per-output device reference and time it.
 *   mode    0 FWD   Y[rows,n_out]  = lrelu(X[rows,k_in] W[n_out,k_in]^T + b)
 *           1 DGRAD dX[rows,k_in]  = (dY[rows,n_out] W[n_out,k_in]) * lrelu'(A[rows,k_in])
 *           2 WGRAD dW[n_out,k_in] = dY^T X ; db = colsum(dY)
 *   variant kernel family / tile (see gemm_bench.hip for the table)
 *   groups  number of independent problems carried by one launch (1..4)
 *   iters   timed back-to-back launches on one stream (after 3 warm-up launches)
 * Returns 0 on success; avg_us = mean time per launch, max_abs_err vs the reference,
 * max_ref = max |reference| (for scaling the error). */
int dqnhip_test_gemm(int32_t mode, int32_t variant, int32_t rows, int32_t n_out, int32_t k_in,
                     int32_t groups, int32_t iters, float* avg_us, float* max_abs_err, float* max_ref);

#ifdef __cplusplus
}
#endif
#endif
