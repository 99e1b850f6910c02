/*
 * dqnhip_internal.h — test / tuning hooks exported by libdqnhip_test.so (csrc/gemm_bench.hip).  Not part of the
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

/* fp16-input / fp32-accumulate GEMM family (csrc/hgemm.hip.h) on random data against a naive device
 * reference.  C[m][n] = sum_k A[m][k] B[n][k], A [M][K], B [N][K] fp16.
 *   mode 0 FWD-like   bias + leaky ReLU; fp16 [M][N], transposed fp16 [N][M] and fp32 outputs
 *        1 DGRAD-like ReLU' mask; the same three outputs, fp32 one scaled
 *        2 WGRAD-like scaled fp32 output, only the first N/2 columns written
 *        4 / 5   as 0 with the fp16 output only / fp16 + transposed fp16 (the common layer cases)
 *        3 glue check: k_cvt16 (fp32 -> fp16 + transposed, zero padding) and k_db16 on an M x N panel
 *   tile 0 auto, 1 128x128, 2 64x64 with in-workgroup split-K, 3 256x128 on eight waves; +10: two problems (the second a
 *        copy with its own outputs, which must come out bit-identical) in one launch
 * max_abs_err excludes one fp16 rounding of each fp16 result. */
int dqnhip_test_hgemm(int32_t mode, int32_t tile, int32_t M, int32_t N, int32_t K, int32_t iters,
                      float* avg_us, float* max_abs_err, float* max_ref);

/* One tower layer's backward without transposed panels (see gemm_bench.hip): dgrad with the weight operand read
 * reduction-major, wgrad with both operands reduction-major, each alone and both in one launch; us[3] = the three
 * timings, max_abs_err over all results against the naive reference. */
int dqnhip_test_hgemm_backward(int32_t rows, int32_t n_out, int32_t k_in, int32_t iters, float* us, float* max_abs_err, float* max_ref);

/* Times the fused clip+Adam+soft-update pass on n_params random parameters (see gemm_bench.hip).
 * variant = 10*U + NT (U in {1,2,4} float4 per array in flight per thread, NT = non-temporal
 * gradient loads); touch_mb = MB of unrelated traffic between two passes (0: back to back). */
int dqnhip_test_adam(int64_t n_params, int32_t variant, int32_t blocks, int32_t iters, int32_t touch_mb, float* avg_us);

/* Persistent-kernel probe: `layers` dependent 256 x 1024 x 1024 forward layers as `layers` launches of the
 * learner's gemm_fwd_lds<2,2> vs ONE launch of 256 co-resident workgroups with per-(layer, 32-row slab)
 * arrival counters (see gemm_bench.hip).  map bit 0: 0 the learner's tile->XCD map, 1 one slab per XCD;
 * bit 1: hand-off by write-through (sc1) tile stores without a release fence instead of plain stores + release;
 * bit 2: the SEPARATE launches use write-through output stores (does a boundary get cheaper with nothing dirty?).
 * max_abs_diff compares the two results (same arithmetic: expected 0); gave_up != 0 if a bounded spin expired. */
int dqnhip_test_chain(int32_t layers, int32_t map, int32_t iters, float* us_launches, float* us_persistent,
                      float* max_abs_diff, int32_t* gave_up);

/* CU load-path probe (see gemm_bench.hip): `blocks` workgroups of 256 threads each stream `iters` 32-KiB pieces from a
 * region of region_kb KiB shared by the workgroups of one XCD.  mode 0 register loads, 1 LDS-DMA, 2 LDS-DMA + fragment
 * reads.  tb_per_s = bytes delivered to the CUs per second, chip-wide. */
int dqnhip_test_loadpath(int32_t mode, int32_t blocks, int32_t region_kb, int32_t iters, int32_t launches,
                         float* avg_us, float* tb_per_s);

/* Optimiser-under-GEMM overlap probe (see gemm_bench.hip): `layers` dependent 256 x 1024 x 1024 forward launches and one
 * k_adam_soft pass over adam_params parameters; us[0] launches alone, [1] pass alone, [2] serial, [3] the pass as rider
 * workgroups of the launches (48-KiB LDS build: riders co-resident), [4] on a second stream, [5] riders with the 96-KiB
 * build (not co-resident).  rider_blocks = rider workgroups per launch. */
int dqnhip_test_overlap(int32_t layers, int64_t adam_params, int32_t rider_blocks, int32_t iters, float* us);
/* launch-floor probe: us per kernel of a `chain`-long dependent chain inside a replayed hipGraph (variant bits: 1 = 640-byte
 * kernarg, 2 = lds_bytes of dynamic LDS, 4 = one dependent global round trip in the body, 8 = 1024 threads per block) */
int dqnhip_test_launch_floor(int32_t variant, int32_t blocks, int32_t lds_bytes, int32_t chain, int32_t iters, float* us_per_kernel);

#ifdef __cplusplus
}
#endif
#endif
