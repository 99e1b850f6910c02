// gemm_bench.hip — dqnhip_test_gemm: correctness + timing harness for the GEMM kernel
// families (tests/csrc/dqnhip_internal.h).  Test/tuning infrastructure, not on the hot path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dqnhip_internal.h"
#include "gemm_direct.hip.h"
#include "gemm_mfma.hip.h"

using namespace dqnhip;

namespace {

__global__ void k_fill(float* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}
__global__ void k_ref_fwd(const float* X, const float* W, const float* b, float* Y, int M, int N, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i % N;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(X[(size_t)m * K + k], W[(size_t)n * K + k], acc);
  Y[i] = lrelu_fwd(acc + b[n]);
}
__global__ void k_ref_dgrad(const float* dY, const float* W, const float* A, float* dX, int M, int N, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * K) return;
  const int m = i / K, k = i % K;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) acc = fmaf(dY[(size_t)m * N + n], W[(size_t)n * K + k], acc);
  dX[i] = acc * lrelu_mask(A[i]);
}
__global__ void k_ref_wgrad(const float* dY, const float* X, float* dW, float* db, int M, int N, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  const int n = i / K, k = i % K;
  float acc = 0.f, bs = 0.f;
  for (int m = 0; m < M; ++m) { acc = fmaf(dY[(size_t)m * N + n], X[(size_t)m * K + k], acc); bs += dY[(size_t)m * N + n]; }
  dW[i] = acc;
  if (k == 0) db[n] = bs;
}
__global__ void k_maxdiff(const float* a, const float* b, size_t n, float* out /*[2]*/) {
  __shared__ float sd[256], sr[256];
  float d = 0.f, r = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) { d = fmaxf(d, fabsf(a[i] - b[i])); r = fmaxf(r, fabsf(b[i])); }
  sd[threadIdx.x] = d; sr[threadIdx.x] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sd[threadIdx.x] = fmaxf(sd[threadIdx.x], sd[threadIdx.x + s]); sr[threadIdx.x] = fmaxf(sr[threadIdx.x], sr[threadIdx.x + s]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = fmaxf(out[0], sd[0]); out[1] = fmaxf(out[1], sr[0]); }
}

// pure MFMA issue-rate probe: NACC independent accumulators, no memory traffic in the loop
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < NACC; ++e) acc[e] = DQN_MFMA(a, b, acc[e]);
  }
  f32x4 r = acc[0];
#pragma unroll
  for (int e = 1; e < NACC; ++e) { r.x += acc[e].x; r.y += acc[e].y; r.z += acc[e].z; r.w += acc[e].w; }
  reinterpret_cast<f32x4*>(out)[blockIdx.x * 256 + threadIdx.x] = r;
}

// the same for v_mfma_f32_32x32x16_f16 (the fp16 learner's instruction): NACC independent 32x32 accumulators
typedef __attribute__((ext_vector_type(8))) _Float16 pk_h16x8;
typedef __attribute__((ext_vector_type(16))) float pk_f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_peak_h(float* out, int iters) {
  pk_f32x16 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[e][j] = 0.f;
  pk_h16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < NACC; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[e], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int e = 0; e < NACC; ++e)
#pragma unroll
    for (int j = 0; j < 16; ++j) r += acc[e][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

#define CK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { fprintf(stderr, "dqnhip_test_gemm: %s -> %s\n", #e, hipGetErrorString(e__)); return 2; } } while (0)

hipError_t launch_variant(int mode, int variant, GemmBatch& b, hipStream_t s) {
  // variant 0: LDS-staged family (gemm_mfma.hip.h); 1..: direct family (gemm_direct.hip.h)
  if (mode == GEMM_FWD) {
    switch (variant) {
      case 0: return gemm_launch<GEMM_FWD, 64, 32, 2, 2>(b, s);
      case 1: return fwd_direct_launch<2, 2>(b, s);   // 32x32
      case 2: return fwd_direct_launch<4, 2>(b, s);   // 64x32
      case 3: return fwd_direct_launch<2, 1>(b, s);   // 32x16
      case 4: return fwd_direct_launch<4, 4>(b, s);   // 64x64
      case 10: return fwd_lds_launch<2, 2, false>(b, s);  // 32x32 coalesced+LDS transpose
      case 11: return fwd_lds_launch<2, 2, true>(b, s);
      case 12: return fwd_lds_launch<4, 2, false>(b, s);  // 64x32
      case 13: return fwd_lds_launch<4, 2, true>(b, s);
      case 14: return fwd_lds_launch<4, 4, false>(b, s);  // 64x64
      case 15: return fwd_lds_launch<1, 1, false>(b, s);  // 16x16 (acting-time batches)
      case 16: return fwd_lds_launch<2, 1, false>(b, s);  // 32 outputs x 16 rows
      case 20: return fwd_lds_launch<1, 1, true>(b, s);   // 16x16, staged instructions interleaved
      case 17: return fwd_lds_launch<4, 2, false, 1>(b, s);  // 64x32, one LDS image per wave (2 workgroups per CU)
      case 18: return fwd_lds_launch<2, 2, false, 1>(b, s);
      case 19: return fwd_lds_launch<4, 4, false, 1>(b, s);
    }
  } else if (mode == GEMM_DGRAD) {
    switch (variant) {
      case 0: return gemm_launch<GEMM_DGRAD, 64, 32, 2, 2>(b, s);
      case 1: return dgrad_direct_launch<1, 1>(b, s);  // 64x16
      case 2: return dgrad_direct_launch<1, 2>(b, s);  // 64x32
      case 3: return dgrad_direct_launch<2, 1>(b, s);  // 128x16
      case 4: return dgrad_direct_launch<1, 4>(b, s);  // 64x64
      case 5: return dgrad_lds_launch<1, 1>(b, s);     // 64x16, dY through the LDS transpose
      case 6: return dgrad_lds_launch<1, 2>(b, s);     // 64x32
      case 7: return dgrad_narrow_launch(b, s);        // 16 input columns x 16 rows
      case 8: return dgrad_direct32_launch(b, s);      // 32 x 16 tiles: twice the workgroups, co-resident (round 6 experiment)
    }
  } else {
    switch (variant) {
      case 0: return gemm_launch<GEMM_WGRAD, 64, 64, 2, 2>(b, s);
      case 1: return wgrad_direct_launch<1, 1>(b, s);  // 64x64
      case 2: return wgrad_narrow_launch<1>(b, s);     // 64 x 16 outputs (first layer)
      case 3: return wgrad_quad_launch<8>(b, s);       // 64x64, one 32x32 quadrant per wave over the whole reduction (no split-K, no LDS pass)
      case 4: return wgrad_quad_launch<4>(b, s);
      case 5: return wgrad_quad_launch<16>(b, s);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace

extern "C" int dqnhip_test_gemm(int32_t mode, int32_t variant, int32_t rows, int32_t n_out, int32_t k_in,
                                int32_t groups, int32_t iters, float* avg_us, float* max_abs_err, float* max_ref) {
  if (mode == 99) {   // MFMA peak probe: variant = accumulators per wave, rows = blocks, n_out = loop iterations
    hipStream_t s; CK(hipStreamCreate(&s));
    float* out; CK(hipMalloc(&out, (size_t)rows * 256 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) {
        if (variant == 11) hipLaunchKernelGGL(k_mfma_peak_h<1>, dim3(rows), dim3(256), 0, s, out, n_out);
        else if (variant == 12) hipLaunchKernelGGL(k_mfma_peak_h<2>, dim3(rows), dim3(256), 0, s, out, n_out);
        else if (variant == 14) hipLaunchKernelGGL(k_mfma_peak_h<4>, dim3(rows), dim3(256), 0, s, out, n_out);
        else if (variant == 1) hipLaunchKernelGGL(k_mfma_peak<1>, dim3(rows), dim3(256), 0, s, out, n_out, 0.5f, 0.25f);
        else if (variant == 2) hipLaunchKernelGGL(k_mfma_peak<2>, dim3(rows), dim3(256), 0, s, out, n_out, 0.5f, 0.25f);
        else hipLaunchKernelGGL(k_mfma_peak<4>, dim3(rows), dim3(256), 0, s, out, n_out, 0.5f, 0.25f);
      }
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us) *avg_us = ms * 1000.0f / iters;
    hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return 0;
  }
  if (groups < 1 || groups > kMaxGroup || iters < 1) return 1;
  static bool prepared = false;
  if (!prepared) {
    CK((gemm_prepare<GEMM_FWD, 64, 32, 2, 2>())); CK((gemm_prepare<GEMM_DGRAD, 64, 32, 2, 2>()));
    CK((gemm_prepare<GEMM_WGRAD, 64, 64, 2, 2>()));
    CK(direct_prepare(gemm_wgrad_direct<1, 1>, 4 * 16 * 64 * 16 + 4 * 16 * 16));
    CK(direct_prepare(gemm_fwd_direct<4, 4>, 4 * 16 * 64 * 16));
    CK(direct_prepare(gemm_fwd_lds<4, 2, false>, 4 * 2 * 6 * 512 * 4)); CK(direct_prepare(gemm_fwd_lds<4, 2, true>, 4 * 2 * 6 * 512 * 4));
    CK(direct_prepare(gemm_fwd_lds<4, 4, false>, 4 * 2 * 8 * 512 * 4));
    CK(direct_prepare((gemm_fwd_lds<4, 4, false, 1>), (fwd_lds_bytes<4, 4, false, 1>())));
    CK(direct_prepare(gemm_dgrad_direct<1, 4>, 4 * 16 * 64 * 16));
    prepared = true;
  }
  hipStream_t s; CK(hipStreamCreate(&s));
  const int M = rows, N = n_out, K = k_in;
  const size_t nX = (size_t)M * K, nW = (size_t)N * K, nY = (size_t)M * N;
  std::vector<float*> X(groups), W(groups), Yv(groups), bias(groups), out(groups), ref(groups), db(groups), dbref(groups), part(groups);
  const size_t nOut = mode == GEMM_FWD ? nY : mode == GEMM_DGRAD ? nX : nW;
  for (int g = 0; g < groups; ++g) {
    CK(hipMalloc(&X[g], nX * 4)); CK(hipMalloc(&W[g], nW * 4)); CK(hipMalloc(&Yv[g], nY * 4));
    CK(hipMalloc(&bias[g], N * 4)); CK(hipMalloc(&out[g], nOut * 4)); CK(hipMalloc(&ref[g], nOut * 4));
    CK(hipMalloc(&db[g], N * 4)); CK(hipMalloc(&dbref[g], N * 4)); CK(hipMalloc(&part[g], 65536 * 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, X[g], nX, 11u + g, -1.f, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, W[g], nW, 23u + g, -1.f, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, Yv[g], nY, 37u + g, -1.f, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, s, bias[g], (size_t)N, 41u + g, -1.f, 1.f);
    CK(hipMemsetAsync(out[g], 0xff, nOut * 4, s));
  }
  GemmBatch b{}; b.n = groups;
  for (int g = 0; g < groups; ++g) {
    GemmProblem& p = b.prob[g];
    if (mode == GEMM_FWD) {
      p.P = W[g]; p.ldp = K; p.Q = X[g]; p.ldq = K; p.C = out[g]; p.ldc = N; p.Pdim = N; p.Qdim = M; p.Kred = K;
      p.bias = bias[g]; p.relu = 1;
    } else if (mode == GEMM_DGRAD) {
      p.P = W[g]; p.ldp = K; p.Q = Yv[g]; p.ldq = N; p.C = out[g]; p.ldc = K; p.Pdim = K; p.Qdim = M; p.Kred = N;
      p.mask = X[g]; p.ldm = K;
    } else {
      p.P = X[g]; p.ldp = K; p.Q = Yv[g]; p.ldq = N; p.C = out[g]; p.ldc = K; p.Pdim = K; p.Qdim = N; p.Kred = M;
      p.db = db[g]; p.partial = part[g];
    }
  }
  for (int i = 0; i < 3; ++i) CK(launch_variant(mode, variant, b, s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (getenv("DQNHIP_BENCH_STREAMS")) {
    // concurrency probe: the same launch sequence on NS independent streams (separate problems
    // per stream would be cleaner; outputs are identical so the write race is benign here)
    const int NS = atoi(getenv("DQNHIP_BENCH_STREAMS"));
    std::vector<hipStream_t> ss(NS);
    std::vector<hipEvent_t> ev(NS);
    for (int k = 0; k < NS; ++k) { CK(hipStreamCreateWithFlags(&ss[k], hipStreamNonBlocking)); CK(hipEventCreate(&ev[k])); }
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int k = 0; k < NS; ++k) CK(hipStreamWaitEvent(ss[k], e0, 0));
      for (int i = 0; i < iters; ++i)
        for (int k = 0; k < NS; ++k) CK(launch_variant(mode, variant, b, ss[k]));
      for (int k = 0; k < NS; ++k) { CK(hipEventRecord(ev[k], ss[k])); CK(hipStreamWaitEvent(s, ev[k], 0)); }
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    }
    float ms2 = 0; CK(hipEventElapsedTime(&ms2, e0, e1));
    if (avg_us) *avg_us = ms2 * 1000.0f / iters;     // per "round" of NS concurrent launches
    for (int k = 0; k < NS; ++k) { hipStreamDestroy(ss[k]); hipEventDestroy(ev[k]); }
    if (max_abs_err) *max_abs_err = 0; if (max_ref) *max_ref = 0;
    return 0;
  }
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(launch_variant(mode, variant, b, s));
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  if (avg_us) *avg_us = ms * 1000.0f / iters;
  // reference + compare
  float* dres; CK(hipMalloc(&dres, 8)); CK(hipMemsetAsync(dres, 0, 8, s));
  for (int g = 0; g < groups; ++g) {
    const int threads = 256;
    if (mode == GEMM_FWD) hipLaunchKernelGGL(k_ref_fwd, dim3((nOut + 255) / 256), dim3(threads), 0, s, X[g], W[g], bias[g], ref[g], M, N, K);
    else if (mode == GEMM_DGRAD) hipLaunchKernelGGL(k_ref_dgrad, dim3((nOut + 255) / 256), dim3(threads), 0, s, Yv[g], W[g], X[g], ref[g], M, N, K);
    else hipLaunchKernelGGL(k_ref_wgrad, dim3((nOut + 255) / 256), dim3(threads), 0, s, Yv[g], X[g], ref[g], dbref[g], M, N, K);
    hipLaunchKernelGGL(k_maxdiff, dim3(1), dim3(256), 0, s, out[g], ref[g], nOut, dres);
    if (mode == GEMM_WGRAD) hipLaunchKernelGGL(k_maxdiff, dim3(1), dim3(256), 0, s, db[g], dbref[g], (size_t)N, dres);
  }
  float hres[2] = {0, 0};
  CK(hipMemcpyAsync(hres, dres, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (max_abs_err) *max_abs_err = hres[0];
  if (max_ref) *max_ref = hres[1];
  for (int g = 0; g < groups; ++g) {
    hipFree(X[g]); hipFree(W[g]); hipFree(Yv[g]); hipFree(bias[g]); hipFree(out[g]); hipFree(ref[g]);
    hipFree(db[g]); hipFree(dbref[g]); hipFree(part[g]);
  }
  hipFree(dres); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  return 0;
}

// ======================= fp16-input family (hgemm.hip.h) ========================================
#define HG_CLOCKPROBE 1
#define HG_WITH_CT16 1      // test build: the transposed-output epilogue / conversion / row-sum kernels the learner no longer uses
#include "hgemm.hip.h"

namespace {

__global__ void k_fill_h(h16* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (h16)(lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f));
  }
}
// naive reference of the whole epilogue, fp32 accumulation of the fp16 inputs in k order
__global__ void k_ref_h(const h16* A, int lda, const h16* B, int ldb, int M, int N, int K, const float* bias, int relu,
                        const h16* mask, int ldm, float* ref, int tn = 0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float acc = 0.f;
  if (tn == 1) { for (int k = 0; k < K; ++k) acc = fmaf((float)A[(size_t)k * lda + m], (float)B[(size_t)k * ldb + n], acc); }
  else if (tn == 2) { for (int k = 0; k < K; ++k) acc = fmaf((float)A[(size_t)m * lda + k], (float)B[(size_t)k * ldb + n], acc); }
  else
  for (int k = 0; k < K; ++k) acc = fmaf((float)A[(size_t)m * lda + k], (float)B[(size_t)n * ldb + k], acc);
  if (bias) acc += bias[n];
  if (relu) acc = acc > 0.f ? acc : 0.01f * acc;
  if (mask) acc *= ((float)mask[(size_t)m * ldm + n] > 0.f ? 1.0f : 0.01f);
  ref[i] = acc;
}
// compare fp32 ref [M][N] with: fp16 m-major, fp16 transposed, fp32 (scaled, first n_valid columns)
__global__ void k_cmp_h(const float* ref, int M, int N, const h16* C16, int ldc16, const h16* CT16, int ldct, const float* C32,
                        int ldc32, int n_valid, float scale, float* out /*[2]: max err, max |ref|*/) {
  __shared__ float sd[256], sr[256];
  float d = 0.f, r = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)M * N; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / N), n = (int)(i % N);
    const float v = ref[i];
    r = fmaxf(r, fabsf(v));
    const float tol16 = fabsf(v) * (1.0f / 1024.0f);          // one fp16 rounding of the result is allowed
    if (C16) d = fmaxf(d, fmaxf(0.f, fabsf((float)C16[(size_t)m * ldc16 + n] - v) - tol16));
    if (CT16) d = fmaxf(d, fmaxf(0.f, fabsf((float)CT16[(size_t)n * ldct + m] - v) - tol16));
    if (C32 && n < n_valid) d = fmaxf(d, fabsf(C32[(size_t)m * ldc32 + n] - v * scale) / fmaxf(scale, 1e-30f));
  }
  sd[threadIdx.x] = d; sr[threadIdx.x] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sd[threadIdx.x] = fmaxf(sd[threadIdx.x], sd[threadIdx.x + s]); sr[threadIdx.x] = fmaxf(sr[threadIdx.x], sr[threadIdx.x + s]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicMax((int*)&out[0], __float_as_int(sd[0])); atomicMax((int*)&out[1], __float_as_int(sr[0])); }
}

}  // namespace

// fp16 GEMM harness.  mode 0: FWD-like (bias + leaky ReLU; fp16, transposed fp16 and fp32 outputs)
//                     mode 1: DGRAD-like (ReLU' mask; fp16 + transposed fp16 + scaled fp32 outputs)
//                     mode 2: WGRAD-like (scaled fp32 output only, the upper half of the columns not written)
//                     mode 3: k_cvt16 + k_db16 glue check (M x N panel)
// tile: 0 auto, 1 force 128x128, 2 force 64x64 split-K, 3 force 256x128 (8 waves); +10: two problems in one launch.
// max_abs_err excludes one fp16 rounding of the result.
extern "C" int dqnhip_test_hgemm(int32_t mode, int32_t tile, int32_t M, int32_t N, int32_t K, int32_t iters,
                                 float* avg_us, float* max_abs_err, float* max_ref) {
  if (iters < 1 || M % 64 || N % 64 || K % 64) return 1;
  static bool prepared = false;
  if (!prepared) { CK(hgemm_prepare_all()); prepared = true; }
  hipStream_t s; CK(hipStreamCreate(&s));
  float* dres; CK(hipMalloc(&dres, 8)); CK(hipMemsetAsync(dres, 0, 8, s));
  float hres[2] = {0, 0};
  if (mode == 3) {
    float* src; h16 *d16, *dT; float *db; 
    const int ld16 = (N + 127) / 128 * 128;
    CK(hipMalloc(&src, (size_t)M * N * 4)); CK(hipMalloc(&d16, (size_t)M * ld16 * 2)); CK(hipMalloc(&dT, (size_t)ld16 * M * 2)); CK(hipMalloc(&db, ld16 * 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, src, (size_t)M * N, 5u, -1.f, 1.f);
    CK(hipMemsetAsync(d16, 0xff, (size_t)M * ld16 * 2, s)); CK(hipMemsetAsync(dT, 0xff, (size_t)ld16 * M * 2, s));
    Cvt16Batch b{}; cvt16_add(b, src, N, M, N - 3, d16, ld16, 2.0f, dT, M);
    CK(cvt16_launch(b, s));
    Db16Batch q{}; q.n = 1; q.scale = 0.5f; q.d[0] = Db16{dT, M, ld16, M, db, 0};
    hipLaunchKernelGGL(k_db16<0>, dim3(ld16), dim3(256), 0, s, q);
    std::vector<float> hs((size_t)M * N), hdb(ld16); std::vector<h16> h16v((size_t)M * ld16), hT((size_t)ld16 * M);
    CK(hipMemcpyAsync(hs.data(), src, hs.size() * 4, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(h16v.data(), d16, h16v.size() * 2, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hT.data(), dT, hT.size() * 2, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hdb.data(), db, hdb.size() * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    float err = 0.f, mx = 0.f;
    for (int c = 0; c < ld16; ++c) {
      double colsum = 0;
      for (int r = 0; r < M; ++r) {
        const float want = c < N - 3 ? (float)(h16)(hs[(size_t)r * N + c] * 2.0f) : 0.f;
        err = fmaxf(err, fabsf((float)h16v[(size_t)r * ld16 + c] - want));
        err = fmaxf(err, fabsf((float)hT[(size_t)c * M + r] - want));
        mx = fmaxf(mx, fabsf(want)); colsum += want;
      }
      err = fmaxf(err, fabsf(hdb[c] - (float)(colsum * 0.5)) / 64.0f);
    }
    if (max_abs_err) *max_abs_err = err; if (max_ref) *max_ref = mx; if (avg_us) *avg_us = 0;
    hipFree(src); hipFree(d16); hipFree(dT); hipFree(db); hipFree(dres); hipStreamDestroy(s);
    return 0;
  }
  h16 *A, *B, *mask, *C16, *CT16; float *bias, *C32, *ref;
  // DQNHIP_TEST_LDPAD=<halves>: leading-dimension padding of both operands (L2 channel-camping probe)
  const int ldpad = getenv("DQNHIP_TEST_LDPAD") ? atoi(getenv("DQNHIP_TEST_LDPAD")) : 0;
  const int ldk = K + ldpad;
  const bool tn = mode == 6;           // mode 6: as 2, but both operands reduction-major ([K][M] and [K][N]) — the learner's wgrad
  const size_t a_elems = tn ? (size_t)K * (M + ldpad) : (size_t)M * ldk, b_elems = tn ? (size_t)K * (N + ldpad) : (size_t)N * ldk;
  CK(hipMalloc(&A, a_elems * 2)); CK(hipMalloc(&B, b_elems * 2)); CK(hipMalloc(&mask, (size_t)M * N * 2));
  CK(hipMalloc(&C16, (size_t)M * N * 2)); CK(hipMalloc(&CT16, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
  CK(hipMalloc(&C32, (size_t)M * N * 4)); CK(hipMalloc(&ref, (size_t)M * N * 4));
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, A, a_elems, 11u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, B, b_elems, 23u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, mask, (size_t)M * N, 31u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, s, bias, (size_t)N, 41u, -1.f, 1.f);
  CK(hipMemsetAsync(C16, 0xff, (size_t)M * N * 2, s)); CK(hipMemsetAsync(CT16, 0xff, (size_t)M * N * 2, s)); CK(hipMemsetAsync(C32, 0xff, (size_t)M * N * 4, s));
  HGemm g{};
  g.A = A; g.lda = ldk; g.B = B; g.ldb = ldk; g.M = M; g.N = N; g.K = K; g.scale32 = 1.0f; g.n_valid32 = N;
  if (mode == 0) { g.bias = bias; g.relu = 1; g.C16 = C16; g.ldc16 = N; g.CT16 = CT16; g.ldct16 = M; g.C32 = C32; g.ldc32 = N; }
  else if (mode == 4) { g.bias = bias; g.relu = 1; g.C16 = C16; g.ldc16 = N; }
  else if (mode == 5) { g.bias = bias; g.relu = 1; g.C16 = C16; g.ldc16 = N; g.CT16 = CT16; g.ldct16 = M; }
  else if (mode == 1) { g.mask = mask; g.ldm = N; g.C16 = C16; g.ldc16 = N; g.CT16 = CT16; g.ldct16 = M; g.C32 = C32; g.ldc32 = N; g.scale32 = 1.0f / 64.0f; }
  else { g.C32 = C32; g.ldc32 = N; g.scale32 = 1.0f / 1024.0f; g.n_valid32 = N / 2; }
  if (tn) { g.ta = g.tb = 1; g.lda = M + ldpad; g.ldb = N + ldpad; }
  // tile >= 10: TWO problems in one launch (the second one a copy with its own outputs), tile - 10 = the forced shape
  const bool pair = tile >= 10;
  if (pair) tile -= 10;
  HGemm gp[2] = {g, g};
  h16 *C16b = nullptr, *CT16b = nullptr; float* C32b = nullptr;
  if (pair) {
    CK(hipMalloc(&C16b, (size_t)M * N * 2)); CK(hipMalloc(&CT16b, (size_t)M * N * 2)); CK(hipMalloc(&C32b, (size_t)M * N * 4));
    CK(hipMemsetAsync(C16b, 0xff, (size_t)M * N * 2, s));
    if (g.C16) gp[1].C16 = C16b; if (g.CT16) gp[1].CT16 = CT16b; if (g.C32) gp[1].C32 = C32b;
  }
  for (int i = 0; i < 3; ++i) CK(hgemm_launch_batch(gp, pair ? 2 : 1, s, tile));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(hgemm_launch_batch(gp, pair ? 2 : 1, s, tile));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  if (avg_us) *avg_us = ms * 1000.0f / iters;
  hipLaunchKernelGGL(k_ref_h, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, s, (const h16*)A, g.lda, (const h16*)B, g.ldb, M, N, K,
                     (const float*)g.bias, g.relu, g.mask, g.ldm, ref, tn ? 1 : 0);
  hipLaunchKernelGGL(k_cmp_h, dim3(256), dim3(256), 0, s, (const float*)ref, M, N, (const h16*)g.C16, N, (const h16*)g.CT16, M, (const float*)g.C32, N,
                     g.n_valid32, g.scale32, dres);
  CK(hipMemcpyAsync(hres, dres, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (mode == 2 || mode == 6) {   // untouched upper columns must still hold the 0xff fill
    std::vector<uint32_t> row(N);
    CK(hipMemcpy(row.data(), C32, N * 4, hipMemcpyDeviceToHost));
    for (int n = N / 2; n < N; ++n) if (row[n] != 0xffffffffu) hres[0] = 1e30f;
  }
  if (getenv("DQNHIP_TEST_CLOCK")) {
    unsigned long long hc[2] = {0, 0};
    CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(hg_clk), sizeof(hc)));
    fprintf(stderr, "hgemm main loop of block 17: %llu shader cycles in %.2f us -> %.0f MHz, %.0f cycles per 64-deep K tile\n", hc[0], hc[1] / 100.0,
            hc[1] ? hc[0] / (hc[1] / 100.0) : 0.0, (double)hc[0] / (K / 64));
  }
  if (pair && g.C16) {   // the second problem of the launch must have produced the same fp16 panel, bit for bit
    std::vector<uint16_t> a((size_t)M * N), b((size_t)M * N);
    CK(hipMemcpy(a.data(), C16, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), C16b, b.size() * 2, hipMemcpyDeviceToHost));
    if (memcmp(a.data(), b.data(), a.size() * 2) != 0) hres[0] = 1e30f;
  }
  if (C16b) { hipFree(C16b); hipFree(CT16b); hipFree(C32b); }
  if (max_abs_err) *max_abs_err = hres[0];
  if (max_ref) *max_ref = hres[1];
  hipFree(A); hipFree(B); hipFree(mask); hipFree(C16); hipFree(CT16); hipFree(bias); hipFree(C32); hipFree(ref); hipFree(dres);
  hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  return 0;
}


// The layer backward as the learner launches it without transposed panels: a dgrad-oriented problem (A = dY [B][N_out]
// k-major, B = W [N_out][K_in] reduction-major; ReLU' mask, fp16 + scaled fp32 outputs) and a wgrad-oriented one (A = dY
// [B][N_out], B = X [B][K_in], both reduction-major; scaled fp32 output), sharing dY.  Each alone, then both in ONE
// launch of the 64x64 split-K tile (hgemm_nt<1,1,2,3>); every result against the naive reference.
//   rows = minibatch B, n_out, k_in (multiples of 128).  us[3] = dgrad alone, wgrad alone, pair.
extern "C" int dqnhip_test_hgemm_backward(int32_t rows, int32_t n_out, int32_t k_in, int32_t iters, float* us, float* max_abs_err, float* max_ref) {
  if (iters < 1 || rows % 128 || n_out % 128 || k_in % 128) return 1;
  static bool prepared = false;
  if (!prepared) { CK(hgemm_prepare_all()); prepared = true; }
  hipStream_t s; CK(hipStreamCreate(&s));
  h16 *dY, *W, *X, *mask, *dX16; float *dX32, *dW32, *refd, *refw, *dres;
  CK(hipMalloc(&dY, (size_t)rows * n_out * 2)); CK(hipMalloc(&W, (size_t)n_out * k_in * 2)); CK(hipMalloc(&X, (size_t)rows * k_in * 2));
  CK(hipMalloc(&mask, (size_t)rows * k_in * 2)); CK(hipMalloc(&dX16, (size_t)rows * k_in * 2)); CK(hipMalloc(&dX32, (size_t)rows * k_in * 4));
  CK(hipMalloc(&dW32, (size_t)n_out * k_in * 4)); CK(hipMalloc(&refd, (size_t)rows * k_in * 4)); CK(hipMalloc(&refw, (size_t)n_out * k_in * 4));
  CK(hipMalloc(&dres, 8));
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, dY, (size_t)rows * n_out, 3u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, W, (size_t)n_out * k_in, 5u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, X, (size_t)rows * k_in, 7u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, s, mask, (size_t)rows * k_in, 9u, -1.f, 1.f);
  HGemm gd{}, gw{};
  gd.A = dY; gd.lda = n_out; gd.B = W; gd.ldb = k_in; gd.tb = 1; gd.M = rows; gd.N = k_in; gd.K = n_out;
  gd.mask = mask; gd.ldm = k_in; gd.C16 = dX16; gd.ldc16 = k_in; gd.C32 = dX32; gd.ldc32 = k_in; gd.n_valid32 = k_in; gd.scale32 = 1.0f / 64.0f;
  gw.A = dY; gw.lda = n_out; gw.ta = 1; gw.B = X; gw.ldb = k_in; gw.tb = 1; gw.M = n_out; gw.N = k_in; gw.K = rows;
  gw.C32 = dW32; gw.ldc32 = k_in; gw.n_valid32 = k_in; gw.scale32 = 1.0f / 1024.0f;
  hipLaunchKernelGGL(k_ref_h, dim3((unsigned)(((size_t)rows * k_in + 255) / 256)), dim3(256), 0, s, (const h16*)dY, n_out, (const h16*)W, k_in, rows, k_in, n_out,
                     (const float*)nullptr, 0, (const h16*)mask, k_in, refd, 2);
  hipLaunchKernelGGL(k_ref_h, dim3((unsigned)(((size_t)n_out * k_in + 255) / 256)), dim3(256), 0, s, (const h16*)dY, n_out, (const h16*)X, k_in, n_out, k_in, rows,
                     (const float*)nullptr, 0, (const h16*)nullptr, 0, refw, 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float worst = 0.f, big = 0.f;
  auto check = [&]() -> int {
    float hres[2];
    CK(hipMemsetAsync(dres, 0, 8, s));
    hipLaunchKernelGGL(k_cmp_h, dim3(256), dim3(256), 0, s, (const float*)refd, rows, k_in, (const h16*)dX16, k_in, (const h16*)nullptr, 0, (const float*)dX32, k_in, k_in, gd.scale32, dres);
    hipLaunchKernelGGL(k_cmp_h, dim3(256), dim3(256), 0, s, (const float*)refw, n_out, k_in, (const h16*)nullptr, 0, (const h16*)nullptr, 0, (const float*)dW32, k_in, k_in, gw.scale32, dres);
    CK(hipMemcpyAsync(hres, dres, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    worst = fmaxf(worst, hres[0]); big = fmaxf(big, hres[1]);
    return 0;
  };
  auto clear = [&]() -> int {
    CK(hipMemsetAsync(dX16, 0xff, (size_t)rows * k_in * 2, s)); CK(hipMemsetAsync(dX32, 0xff, (size_t)rows * k_in * 4, s));
    CK(hipMemsetAsync(dW32, 0xff, (size_t)n_out * k_in * 4, s));
    return 0;
  };
  for (int form = 0; form < 3; ++form) {        // 0: dgrad alone (+ wgrad once for the check), 1: wgrad alone, 2: the pair
    if (clear()) return 2;
    const HGemm pair[2] = {gd, gw};
    auto run = [&]() -> hipError_t {
      if (form == 0) return hgemm_launch(gd, s);
      if (form == 1) return hgemm_launch(gw, s);
      return hgemm_launch_batch(pair, 2, s, 2);
    };
    for (int i = 0; i < 3; ++i) CK(run());
    CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK(run());
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (us) us[form] = ms * 1000.f / iters;
    if (form == 0) CK(hgemm_launch(gw, s));
    if (form == 1) CK(hgemm_launch(gd, s));
    if (check()) return 2;
  }
  if (max_abs_err) *max_abs_err = worst;
  if (max_ref) *max_ref = big;
  hipFree(dY); hipFree(W); hipFree(X); hipFree(mask); hipFree(dX16); hipFree(dX32); hipFree(dW32); hipFree(refd); hipFree(refw); hipFree(dres);
  hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  return 0;
}

// ======================= optimiser pass probe ===================================================
#include "small_kernels.hip.h"

namespace {
__global__ void k_touch(float* p, size_t n4) {        // stand-in for the traffic between two Adam passes of an update
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<f32x4*>(p)[i]; v.x += 1.0f; reinterpret_cast<f32x4*>(p)[i] = v;
  }
}
}  // namespace

// Times the fused clip+Adam+soft-update pass (k_adam_soft) on n_params random parameters.
//   variant  10*U + NT: U = float4s in flight per array per thread (1, 2, 4), NT = non-temporal gradient loads
//   blocks   grid size;  touch_mb: MB of unrelated read-modify-write traffic issued between two passes
//            (0: back-to-back passes over an Infinity-Cache-resident working set)
// avg_us = mean kernel time from the dispatch packets' own timestamps.
extern "C" int dqnhip_test_adam(int64_t n_params, int32_t variant, int32_t blocks, int32_t iters, int32_t touch_mb, float* avg_us) {
  if (n_params < 1024 || n_params % 4 || iters < 1 || blocks < 1) return 1;
  hipStream_t s; CK(hipStreamCreate(&s));
  float *w, *g, *m, *v, *wt, *part, *junk = nullptr;
  DevState* st;
  const size_t nb = (size_t)n_params * 4;
  CK(hipMalloc(&w, nb)); CK(hipMalloc(&g, nb)); CK(hipMalloc(&m, nb)); CK(hipMalloc(&v, nb)); CK(hipMalloc(&wt, nb));
  CK(hipMalloc(&part, 1024 * 4)); CK(hipMalloc(&st, sizeof(DevState)));
  CK(hipMemsetAsync(st, 0, sizeof(DevState), s)); CK(hipMemsetAsync(part, 0, 1024 * 4, s));
  CK(hipMemsetAsync(m, 0, nb, s)); CK(hipMemsetAsync(v, 0, nb, s));
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, w, (size_t)n_params, 1u, -0.1f, 0.1f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, wt, (size_t)n_params, 2u, -0.1f, 0.1f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, g, (size_t)n_params, 3u, -1e-3f, 1e-3f);
  const size_t junk4 = (size_t)touch_mb * (1 << 20) / 16;
  if (touch_mb > 0) { CK(hipMalloc(&junk, junk4 * 16)); CK(hipMemsetAsync(junk, 0, junk4 * 16, s)); }
  AdamArgs a{};
  a.w = w; a.g = g; a.m = m; a.v = v; a.wt = wt; a.n4 = (size_t)n_params / 4; a.partial = part; a.n_partial = 1024;
  a.lr = 1e-3f; a.beta1 = .95f; a.beta2 = .999f; a.eps = 1e-8f; a.clip = 10.f; a.tau = .001f; a.soft_update_freq = 1; a.which = 1; a.st = st;
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  auto launch = [&](hipEvent_t e0, hipEvent_t e1) {
    switch (variant) {
      case 10: hipExtLaunchKernelGGL((k_adam_soft_t<1, 0>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      case 11: hipExtLaunchKernelGGL((k_adam_soft_t<1, 1>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      case 20: hipExtLaunchKernelGGL((k_adam_soft_t<2, 0>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      case 21: hipExtLaunchKernelGGL((k_adam_soft_t<2, 1>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      case 40: hipExtLaunchKernelGGL((k_adam_soft_t<4, 0>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      case 41: hipExtLaunchKernelGGL((k_adam_soft_t<4, 1>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a); break;
      default: return 1;
    }
    return 0;
  };
  for (int i = 0; i < 3; ++i) { if (launch(nullptr, nullptr)) return 1; }
  for (int i = 0; i < iters; ++i) {
    if (touch_mb > 0) hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, s, junk, junk4);
    if (launch(ev[2 * i], ev[2 * i + 1])) return 1;
  }
  CK(hipStreamSynchronize(s));
  double tot = 0;
  for (int i = 0; i < iters; ++i) { float ms = 0; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
  if (avg_us) *avg_us = (float)(tot * 1000.0 / iters);
  for (auto& e : ev) hipEventDestroy(e);
  hipFree(w); hipFree(g); hipFree(m); hipFree(v); hipFree(wt); hipFree(part); hipFree(st); if (junk) hipFree(junk);
  hipStreamDestroy(s);
  return 0;
}


// ======================= persistent forward chain probe ==========================================
// VERDICT r1 item 2 asked for "one persistent kernel per phase with per-row-tile ready flags".  This probe
// measures exactly that seam on the learner's own forward kernel: L dependent 256 x 1024 x 1024 layers
//   (a) as L launches of gemm_fwd_lds<2,2> (what the learner does), and
//   (b) as ONE launch of 256 co-resident workgroups that run the same fwd_lds_body per layer and hand the
//       32-row slabs over through per-(layer, slab) arrival counters: producer plain stores ->
//       s_waitcnt vmcnt(0) -> __syncthreads -> lane-0 agent release -> relaxed counter add; consumer: one lane
//       polls (relaxed, agent) -> agent acquire -> __syncthreads -> plain loads (guide G16's valid form).
//       map = 0: the learner's tile map (a slab's 32 column tiles spread over the 8 XCDs);
//       map = 1: slab s on XCD s (hand-offs stay inside one XCD's L2, every XCD streams all of W).
// Spins are bounded: a give-up sets err and the results are wrong, but nothing can hang.
namespace {
struct ChainArgs {
  const float* W[8]; const float* bias[8]; float* act[9];     // act[0] = input, act[l+1] = output of layer l
  int L, rows, width;
  int* counters;                                               // [L][rows/32], monotone: target = 32 * epoch
  int epoch, map;
  int wt;                                                      // 1: sc1 write-through tile stores, no release fence (guide G16 R1)
  int* err;
};

// the learner's single-layer forward kernel with write-through (sc1) output stores: does a kernel boundary get cheaper
// when the launch leaves no dirty lines behind for the end-of-kernel write-back?  (map bit 2 of dqnhip_test_chain)
__global__ __launch_bounds__(256) void k_fwd_lds_wt(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  fwd_lds_body<2, 2, true, 2, true>(batch.prob[pi], tile_p, tile_q, smem);
}
// ---- round 5: the cheapest in-launch seam this chip offers, for the record ----------------------------------------------
// k_fwd_chain pays, per layer, an agent-scope ACQUIRE on the consumer (~1.7 us: it invalidates the CU's L1) and starts BOTH
// operand streams only after the wait.  Weights do not depend on the chain and an L2 line does not survive a kernel boundary
// (profiles/r05_weight_prefetch.txt) — but it does survive INSIDE a launch.  So (map bit 3): the consumer requests the first
// two 32-k steps of its WEIGHT tile before it polls, takes the activations through device-scope (sc1) buffer loads — which the
// guide allows in place of the acquire when the producer stored sc1 (write-through; map bit 1 is implied) — and no fence
// executes on either side.  Same tile map, same arithmetic, same reduction order as fwd_lds_body<2,2,true,2> (bit-identical).
typedef unsigned chain_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 chain_ld_sc1(const __amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ void fwd_chain_body_ra(const GemmProblem& pr, int tile_p, int tile_q, float* smem, const int* flag, int target,
                                                  int* err, int* s_fail) {
  constexpr int TP = 2, TQ = 2, NB = TP + TQ, NACC = TP * TQ, SLOT = NB * 512, NSLOT = 2;
  constexpr int WSTR = (NSLOT * SLOT > NACC * 256) ? NSLOT * SLOT : NACC * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4, lr = lane >> 3, lc = lane & 7;
  const int p0 = tile_p * 16 * TP, q0 = tile_q * 16 * TQ;
  const int Kw = pr.Kred >> 2, T = Kw >> 5;
  float* wsm = smem + wave * WSTR;
  const float* gp[TP]; int qoff[TQ];
  const size_t ldp8 = (size_t)8 * pr.ldp; const int ldq8b = 8 * pr.ldq * 4;
#pragma unroll
  for (int b = 0; b < TP; ++b) gp[b] = pr.P + (size_t)(p0 + b * 16 + lr) * pr.ldp + wave * Kw + lc * 4;
#pragma unroll
  for (int a = 0; a < TQ; ++a) qoff[a] = ((q0 + a * 16 + lr) * pr.ldq + wave * Kw + lc * 4) * 4;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.Q), 0, pr.Qdim * pr.ldq * 4, 0x00020000);
  const int woff = lr * 32 + ((lc ^ lr) << 2);
  int roff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) roff[kb] = li * 32 + ((((kb << 2) + lg) ^ (li & 7)) << 2);
  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 G0[NB][2], G1[NB][2], F[NB][2], Fn[NB][2];
#define C_GLOADP(G, t) { _Pragma("unroll") for (int b = 0; b < TP; ++b) { \
      G[b][0] = *reinterpret_cast<const f32x4*>(gp[b] + ((t) << 5)); G[b][1] = *reinterpret_cast<const f32x4*>(gp[b] + ldp8 + ((t) << 5)); } }
#define C_GLOADQ(G, t) { _Pragma("unroll") for (int a = 0; a < TQ; ++a) { \
      G[TP + a][0] = chain_ld_sc1(rq, qoff[a] + ((t) << 7)); G[TP + a][1] = chain_ld_sc1(rq, qoff[a] + ldq8b + ((t) << 7)); } }
#define C_GLOAD(G, t) { C_GLOADP(G, t) C_GLOADQ(G, t) }
#define C_SWRITE(slot, G) { _Pragma("unroll") for (int b = 0; b < NB; ++b) { \
      *reinterpret_cast<f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + woff) = G[b][0]; \
      *reinterpret_cast<f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + 256 + woff) = G[b][1]; } }
#define C_SREAD(FF, slot) { _Pragma("unroll") for (int b = 0; b < NB; ++b) { \
      FF[b][0] = *reinterpret_cast<const f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + roff[0]); \
      FF[b][1] = *reinterpret_cast<const f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + roff[1]); } }
#define C_MFMA(FF) { _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) _Pragma("unroll") for (int s = 0; s < 4; ++s) \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a) _Pragma("unroll") for (int c = 0; c < TP; ++c) \
        acc[a * TP + c] = DQN_MFMA(FF[c][kb][s], FF[TP + a][kb][s], acc[a * TP + c]); }
  // the weight stream runs ahead of the dependency
  C_GLOADP(G0, 0) DQN_PIN(); C_GLOADP(G1, 1) DQN_PIN();
  if (flag != nullptr) {
    if (threadIdx.x == 0) {
      int spins = 0, fail = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 400000) { fail = 1; break; }
      }
      *s_fail = fail;
      if (fail) *err = 1;
    }
    __syncthreads();
  }
  C_GLOADQ(G0, 0) DQN_PIN(); C_GLOADQ(G1, 1) DQN_PIN();
  C_SWRITE(0, G0) DQN_PIN(); C_GLOAD(G0, 2) DQN_PIN(); C_SREAD(F, 0) DQN_PIN();
  int t = 0;
  for (; t + 4 < T; t += 2) {
    C_SWRITE(1, G1) DQN_PIN(); C_GLOAD(G1, t + 3) DQN_PIN(); C_SREAD(Fn, 1) DQN_PIN();
    C_MFMA(F) DQN_PIN();
    C_SWRITE(0, G0) DQN_PIN(); C_GLOAD(G0, t + 4) DQN_PIN(); C_SREAD(F, 0) DQN_PIN();
    C_MFMA(Fn) DQN_PIN();
  }
  C_SWRITE(1, G1) C_GLOAD(G1, T - 1) C_SREAD(Fn, 1) DQN_PIN();
  C_MFMA(F) DQN_PIN();
  C_SWRITE(0, G0) C_SREAD(F, 0) DQN_PIN();
  C_MFMA(Fn) DQN_PIN();
  C_SWRITE(1, G1) C_SREAD(Fn, 1) DQN_PIN();
  C_MFMA(F) DQN_PIN();
  C_MFMA(Fn)
#undef C_GLOADP
#undef C_GLOADQ
#undef C_GLOAD
#undef C_SWRITE
#undef C_SREAD
#undef C_MFMA
  constexpr int NBV = (NACC + 3) / 4;
  f32x4 bvp[NBV];
#pragma unroll
  for (int j = 0; j < NBV; ++j) { const int e = j * 4 + wave; if (e < NACC) bvp[j] = *reinterpret_cast<const f32x4*>(pr.bias + p0 + (e % TP) * 16 + (lg << 2)); }
  f32x4* park = reinterpret_cast<f32x4*>(wsm);
#pragma unroll
  for (int e = 0; e < NACC; ++e) park[e * 64 + lane] = acc[e];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pr.C, 0, pr.Qdim * pr.ldc * 4, 0x00020000);
#pragma unroll
  for (int e = 0; e < NACC; ++e) {
    if ((e & 3) == wave) {
      const int a = e / TP, c = e % TP;
      const f32x4 a0 = reinterpret_cast<const f32x4*>(smem + 0 * WSTR)[e * 64 + lane], a1 = reinterpret_cast<const f32x4*>(smem + 1 * WSTR)[e * 64 + lane];
      const f32x4 a2 = reinterpret_cast<const f32x4*>(smem + 2 * WSTR)[e * 64 + lane], a3 = reinterpret_cast<const f32x4*>(smem + 3 * WSTR)[e * 64 + lane];
      f32x4 v;
      v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
      v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
      const int q = q0 + a * 16 + li, p = p0 + c * 16 + (lg << 2);
      const f32x4 bv = bvp[e >> 2];
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      v.x = lrelu_fwd(v.x); v.y = lrelu_fwd(v.y); v.z = lrelu_fwd(v.z); v.w = lrelu_fwd(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(chain_u32x4, v), rs, (int)(((size_t)q * pr.ldc + p) * 4), 0, 16 /* sc1 */);
    }
  }
  __syncthreads();             // the parked tiles are read: the next layer's staging may overwrite them
}
__global__ __launch_bounds__(256) void k_fwd_chain_ra(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_fail;
  const int slabs = a.rows / 32, ctiles = a.width / 32;
  int tile_p, tile_q;
  if (a.map == 0) { const int b = blockIdx.x, xcd = b & 7, j = b >> 3; tile_q = j % slabs; tile_p = (j / slabs) * 8 + xcd; }
  else { const int b = blockIdx.x; tile_q = b & 7; tile_p = b >> 3; }
  if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  for (int l = 0; l < a.L; ++l) {
    GemmProblem pr{};
    pr.P = a.W[l]; pr.ldp = a.width; pr.Q = a.act[l]; pr.ldq = a.width; pr.C = a.act[l + 1]; pr.ldc = a.width;
    pr.Pdim = a.width; pr.Qdim = a.rows; pr.Kred = a.width; pr.bias = a.bias[l]; pr.relu = 1;
    fwd_chain_body_ra(pr, tile_p, tile_q, smem, l > 0 ? a.counters + (l - 1) * slabs + tile_q : nullptr, ctiles * a.epoch, a.err, &s_fail);
    if (l + 1 < a.L) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its write-through stores
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counters + l * slabs + tile_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ __launch_bounds__(256) void k_fwd_chain(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_fail;
  const int slabs = a.rows / 32, ctiles = a.width / 32;
  int tile_p, tile_q;
  if (a.map == 0) { const int b = blockIdx.x, xcd = b & 7, j = b >> 3; tile_q = j % slabs; tile_p = (j / slabs) * 8 + xcd; }
  else { const int b = blockIdx.x; tile_q = b & 7; tile_p = b >> 3; }       // rows/32 == 8 slabs <-> 8 XCDs
  if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // an earlier launch gave up: do nothing
  for (int l = 0; l < a.L; ++l) {
    if (l > 0) {
      if (threadIdx.x == 0) {
        const int* c = a.counters + (l - 1) * slabs + tile_q;
        const int target = ctiles * a.epoch;
        int spins = 0, fail = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 400000) { fail = 1; break; }
          if ((spins & 1023) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { fail = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_fail = fail;
        if (fail) *a.err = 1;
      }
      __syncthreads();
    }
    GemmProblem pr{};
    pr.P = a.W[l]; pr.ldp = a.width; pr.Q = a.act[l]; pr.ldq = a.width; pr.C = a.act[l + 1]; pr.ldc = a.width;
    pr.Pdim = a.width; pr.Qdim = a.rows; pr.Kred = a.width; pr.bias = a.bias[l]; pr.relu = 1;
    if (a.wt) fwd_lds_body<2, 2, true, 2, true>(pr, tile_p, tile_q, smem);
    else fwd_lds_body<2, 2, true>(pr, tile_p, tile_q, smem);
    if (l + 1 < a.L) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its stores
      __syncthreads();
      if (threadIdx.x == 0) {
        if (!a.wt) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(a.counters + l * slabs + tile_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
}  // namespace

extern "C" int dqnhip_test_chain(int32_t layers, int32_t map, int32_t iters, float* us_launches, float* us_persistent,
                                 float* max_abs_diff, int32_t* gave_up) {
  // map bit 0: tile -> XCD map (0 learner's, 1 slab per XCD); bit 1: write-through (sc1) hand-off without a release fence
  const int wt = (map >> 1) & 1, wt_launches = (map >> 2) & 1, run_ahead = (map >> 3) & 1; map &= 1;
  if (layers < 1 || layers > 8 || iters < 1) return 1;
  const int rows = 256, width = 1024;
  hipStream_t s; CK(hipStreamCreate(&s));
  ChainArgs a{}; a.L = layers; a.rows = rows; a.width = width; a.map = map; a.wt = wt;
  float* ref[9];
  for (int l = 0; l < layers; ++l) {
    float *w, *b;
    CK(hipMalloc(&w, (size_t)width * width * 4)); CK(hipMalloc(&b, width * 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, w, (size_t)width * width, 100u + l, -0.05f, 0.05f);
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, s, b, (size_t)width, 200u + l, -0.1f, 0.1f);
    a.W[l] = w; a.bias[l] = b;
  }
  for (int l = 0; l <= layers; ++l) { CK(hipMalloc(&a.act[l], (size_t)rows * width * 4)); CK(hipMalloc(&ref[l], (size_t)rows * width * 4)); }
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, a.act[0], (size_t)rows * width, 7u, -1.f, 1.f);
  CK(hipMemcpyAsync(ref[0], a.act[0], (size_t)rows * width * 4, hipMemcpyDeviceToDevice, s));
  CK(hipMalloc(&a.counters, 8 * 8 * sizeof(int))); CK(hipMemsetAsync(a.counters, 0, 8 * 8 * sizeof(int), s));
  CK(hipMalloc(&a.err, sizeof(int))); CK(hipMemsetAsync(a.err, 0, sizeof(int), s));
  const int lds = fwd_lds_bytes<2, 2, true>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_chain), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_chain_ra), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_lds_wt), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  auto run_launches = [&]() -> hipError_t {
    for (int l = 0; l < layers; ++l) {
      GemmBatch b{}; b.n = 1;
      GemmProblem& p = b.prob[0];
      p.P = a.W[l]; p.ldp = width; p.Q = ref[l]; p.ldq = width; p.C = ref[l + 1]; p.ldc = width;
      p.Pdim = width; p.Qdim = rows; p.Kred = width; p.bias = a.bias[l]; p.relu = 1;
      hipError_t e = wt_launches ? direct_launch(k_fwd_lds_wt, b, 32, 32, (fwd_lds_bytes<2, 2, true>()), s) : fwd_lds_launch<2, 2, true>(b, s);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  int epoch = 0;
  auto run_persistent = [&]() -> hipError_t {
    a.epoch = ++epoch;
    if (run_ahead) hipLaunchKernelGGL(k_fwd_chain_ra, dim3(256), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_fwd_chain, dim3(256), dim3(256), lds, s, a);
    return hipGetLastError();
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int i = 0; i < 3; ++i) CK(run_launches());
  CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(run_launches());
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  if (us_launches) *us_launches = ms * 1000.f / iters;
  for (int i = 0; i < 3; ++i) CK(run_persistent());
  CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(run_persistent());
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  if (us_persistent) *us_persistent = ms * 1000.f / iters;
  float* dres; CK(hipMalloc(&dres, 8)); CK(hipMemsetAsync(dres, 0, 8, s));
  hipLaunchKernelGGL(k_maxdiff, dim3(1), dim3(256), 0, s, a.act[layers], ref[layers], (size_t)rows * width, dres);
  float hres[2] = {0, 0}; int herr = 0;
  CK(hipMemcpyAsync(hres, dres, 8, hipMemcpyDeviceToHost, s));
  CK(hipMemcpyAsync(&herr, a.err, sizeof(int), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (max_abs_diff) *max_abs_diff = hres[0];
  if (gave_up) *gave_up = herr;
  for (int l = 0; l < layers; ++l) { hipFree((void*)a.W[l]); hipFree((void*)a.bias[l]); }
  for (int l = 0; l <= layers; ++l) { hipFree(a.act[l]); hipFree(ref[l]); }
  hipFree(a.counters); hipFree(a.err); hipFree(dres); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  return 0;
}

// ======================= optimiser-under-GEMM overlap probe (VERDICT r3 "Next" item 5) ============================
// Can the HBM / Infinity-Cache-bound optimiser pass hide under MFMA-bound launches?  L dependent 256 x 1024 x 1024 forward
// layers (the learner's gemm_fwd_lds<2,2>) and ONE k_adam_soft pass over adam_params parameters, timed
//   us[0]  the L launches alone            us[1]  the optimiser pass alone (1536 blocks)
//   us[2]  serial on one stream: pass, then the L launches (what the update does today)
//   us[3]  RIDERS: launch l carries slice l of the pass as extra workgroups behind its 256 GEMM tiles.  One LDS image per
//          wave (48 KiB) so that a rider workgroup fits on a CU beside a GEMM workgroup (with the 96-KiB build the
//          riders of a launch cannot become resident before a GEMM workgroup retires: they would run AFTER it)
//   us[4]  TWO STREAMS: the pass on a second stream beside the L launches (fork / join by events, one pair per iteration)
//   us[5]  as us[3] with the 96-KiB build (the non-co-resident control)
// All per iteration, wall time by events on the main stream.
namespace {
template <int NSLOT>
__global__ __launch_bounds__(256) void k_fwd_lds_adam(const GemmBatch batch, AdamArgs a, int adam_blocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < batch.total_tiles) {
    int pi, tile_p, tile_q;
    tile_of_block(batch, pi, tile_p, tile_q);
    fwd_lds_body<2, 2, true, NSLOT>(batch.prob[pi], tile_p, tile_q, smem);
    return;
  }
  adam_soft_body<1, 0>(a, (int)blockIdx.x - batch.total_tiles, adam_blocks, smem);
}
}  // namespace

extern "C" int dqnhip_test_overlap(int32_t layers, int64_t adam_params, int32_t rider_blocks, int32_t iters, float* us /*[6]*/) {
  if (layers < 1 || layers > 8 || iters < 1 || adam_params < 4096 || adam_params % (4 * layers) || rider_blocks < 1) return 1;
  const int rows = 256, width = 1024;
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *W[8], *bias[8], *act[9];
  for (int l = 0; l < layers; ++l) {
    CK(hipMalloc(&W[l], (size_t)width * width * 4)); CK(hipMalloc(&bias[l], width * 4));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, W[l], (size_t)width * width, 100u + l, -0.05f, 0.05f);
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, s, bias[l], (size_t)width, 200u + l, -0.1f, 0.1f);
  }
  for (int l = 0; l <= layers; ++l) CK(hipMalloc(&act[l], (size_t)rows * width * 4));
  hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, s, act[0], (size_t)rows * width, 7u, -1.f, 1.f);
  float *w, *g, *m, *v, *wt, *part; DevState* st;
  const size_t nb = (size_t)adam_params * 4;
  CK(hipMalloc(&w, nb)); CK(hipMalloc(&g, nb)); CK(hipMalloc(&m, nb)); CK(hipMalloc(&v, nb)); CK(hipMalloc(&wt, nb));
  CK(hipMalloc(&part, 1024 * 4)); CK(hipMalloc(&st, sizeof(DevState)));
  CK(hipMemsetAsync(st, 0, sizeof(DevState), s)); CK(hipMemsetAsync(part, 0, 1024 * 4, s));
  CK(hipMemsetAsync(m, 0, nb, s)); CK(hipMemsetAsync(v, 0, nb, s));
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, w, (size_t)adam_params, 1u, -0.1f, 0.1f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, wt, (size_t)adam_params, 2u, -0.1f, 0.1f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, g, (size_t)adam_params, 3u, -1e-3f, 1e-3f);
  AdamArgs a{};
  a.w = w; a.g = g; a.m = m; a.v = v; a.wt = wt; a.n4 = (size_t)adam_params / 4; a.partial = part; a.n_partial = 1024;
  a.lr = 1e-3f; a.beta1 = .95f; a.beta2 = .999f; a.eps = 1e-8f; a.clip = 10.f; a.tau = .001f; a.soft_update_freq = 1; a.which = 1; a.st = st;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_lds_adam<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (fwd_lds_bytes<2, 2, true, 1>())));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fwd_lds_adam<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (fwd_lds_bytes<2, 2, true, 2>())));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fwd_lds<2, 2, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (fwd_lds_bytes<2, 2, true, 1>())));
  auto problem = [&](int l) { GemmBatch b{}; b.n = 1; GemmProblem& p = b.prob[0];
    p.P = W[l]; p.ldp = width; p.Q = act[l]; p.ldq = width; p.C = act[l + 1]; p.ldc = width;
    p.Pdim = width; p.Qdim = rows; p.Kred = width; p.bias = bias[l]; p.relu = 1;
    p.tiles_p = width / 32; p.tiles_q = rows / 32; p.tile_base = 0; b.total_tiles = p.tiles_p * p.tiles_q; return b; };
  auto chain = [&](hipStream_t st_) -> hipError_t {
    for (int l = 0; l < layers; ++l) { GemmBatch b = problem(l); hipError_t e = fwd_lds_launch<2, 2, true>(b, st_); if (e != hipSuccess) return e; }
    return hipSuccess; };
  auto adam = [&](hipStream_t st_) -> hipError_t { hipLaunchKernelGGL(k_adam_soft, dim3(1536), dim3(256), 0, st_, a); return hipGetLastError(); };
  auto riders = [&](int nslot) -> hipError_t {
    const size_t slice4 = a.n4 / layers;
    for (int l = 0; l < layers; ++l) {
      GemmBatch b = problem(l);
      AdamArgs al = a;
      al.w = w + 4 * slice4 * l; al.g = g + 4 * slice4 * l; al.m = m + 4 * slice4 * l; al.v = v + 4 * slice4 * l; al.wt = wt + 4 * slice4 * l; al.n4 = slice4;
      if (nslot == 1) hipLaunchKernelGGL(k_fwd_lds_adam<1>, dim3(b.total_tiles + rider_blocks), dim3(256), (fwd_lds_bytes<2, 2, true, 1>()), s, b, al, rider_blocks);
      else hipLaunchKernelGGL(k_fwd_lds_adam<2>, dim3(b.total_tiles + rider_blocks), dim3(256), (fwd_lds_bytes<2, 2, true, 2>()), s, b, al, rider_blocks);
      hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    return hipSuccess; };
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  auto timeit = [&](int which, float* out) -> int {
    auto one = [&]() -> hipError_t {
      switch (which) {
        case 0: return chain(s);
        case 1: return adam(s);
        case 2: { hipError_t e = adam(s); return e != hipSuccess ? e : chain(s); }
        case 3: return riders(1);
        case 4: { hipError_t e = hipEventRecord(ef, s); if (e != hipSuccess) return e;
                  e = hipStreamWaitEvent(s2, ef, 0); if (e != hipSuccess) return e;
                  e = adam(s2); if (e != hipSuccess) return e;
                  e = chain(s); if (e != hipSuccess) return e;
                  e = hipEventRecord(ej, s2); if (e != hipSuccess) return e;
                  return hipStreamWaitEvent(s, ej, 0); }
        default: return riders(2);
      } };
    for (int i = 0; i < 5; ++i) CK(one());
    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CK(one());
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    *out = ms * 1000.f / iters;
    return 0; };
  for (int which = 0; which < 6; ++which) { int rc = timeit(which, &us[which]); if (rc) return rc; }
  for (int l = 0; l < layers; ++l) { hipFree(W[l]); hipFree(bias[l]); }
  for (int l = 0; l <= layers; ++l) hipFree(act[l]);
  hipFree(w); hipFree(g); hipFree(m); hipFree(v); hipFree(wt); hipFree(part); hipFree(st);
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(ef); hipEventDestroy(ej); hipStreamDestroy(s); hipStreamDestroy(s2);
  return 0;
}

// ---- CU load-path probe: how many bytes per clock one CU can pull from its XCD's L2 ------------------------------------
// Every workgroup streams `iters` 32-KiB pieces (8 x 1 KiB per wave, 16 B per lane — the fp16 GEMM's stage shape) from a
// region that all workgroups of its XCD share (region_kb per XCD: <= 2 MiB stays L2-resident, 16 KiB stays in the L1).
//   mode 0  global_load_dwordx4 into registers (8 in flight per wave)
//   mode 1  global_load_lds_dwordx4 (LDS-DMA) into a 4-stage ring, counted vmcnt, no consumer
//   mode 2  as 1, plus every wave reads 16 KiB of each landed stage with ds_read_b128 (the GEMM's fragment traffic)
//   mode 3  the fragment reads alone (no DMA)
//   mode 4  global -> registers -> ds_write_b128 (two register sets in flight) + the fragment reads;  5: without the reads
//   mode 6  half of each stage by LDS-DMA, half straight into registers, + the fragment reads;  7: without the reads
//   mode 8  a fifth wave issues all the LDS-DMA of a stage, waves 0-3 only read fragments (loader / consumer split)
//   mode 9  fragment reads alone with the GEMM's swizzled fragment addresses
namespace {
template <int MODE>
__global__ __launch_bounds__(MODE == 8 ? 320 : 256, 1) void k_loadpath(const unsigned char* src, uint32_t region_bytes, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lp_smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned char* base = src + (size_t)xcd * region_bytes;
  const uint32_t npieces = region_bytes / 32768u;               // 32-KiB pieces in the region
  uint32_t piece = (uint32_t)(slot * 7) % npieces;
  uint32_t acc = 0;
  if constexpr (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)piece * 32768u + w * 8192 + lane * 16);
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = p[i * 64];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
      piece = piece + 1 == npieces ? 0 : piece + 1;
    }
  } else if constexpr (MODE == 4 || MODE == 5) {
    // global -> registers -> ds_write_b128, two register sets (two stages of loads in flight), 4 LDS slots
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lp_smem;
    uint4 r0[8], r1[8];
    auto gload = [&](uint4 (&r)[8]) {
      const uint4* p = reinterpret_cast<const uint4*>(base + (size_t)piece * 32768u + w * 8192 + lane * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = p[i * 64];
      piece = piece + 1 == npieces ? 0 : piece + 1;
    };
    auto swrite = [&](const uint4 (&r)[8], int it) {
      uint4* d = reinterpret_cast<uint4*>(lp_smem + (it & 3) * 32768 + w * 8192) + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i * 64] = r[i];
    };
    auto sread = [&](int it) {
      if constexpr (MODE == 4) {
        const uint32_t a0 = lds0 + (uint32_t)((it & 3) * 32768 + (w & 1) * 16384 + lane * 16);
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a0), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 4) acc ^= v[i].x;
      }
    };
    gload(r0); gload(r1);
    swrite(r0, 0); gload(r0);
    for (int it = 0; it < iters; it += 2) {
      __syncthreads();
      swrite(r1, it + 1); gload(r1);
      sread(it);
      __syncthreads();
      swrite(r0, it + 2); gload(r0);
      sread(it + 1);
    }
    acc ^= r0[0].x ^ r1[0].x;
  } else if constexpr (MODE == 8) {
    // loader wave: wave 4 (of 5) issues every LDS-DMA piece of a stage (32 x 1 KiB) and owns the counted vmcnt;
    // waves 0-3 only read fragments.  One barrier per stage, as in the GEMM.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lp_smem;
    const uint32_t voff = lane * 16;
    auto issue = [&](int it) {
      const unsigned char* b = base + (size_t)piece * 32768u;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const uint32_t dst = lds0 + (uint32_t)((it & 3) * 32768 + i * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(b + i * 1024), "s"(dst) : "memory");
      }
      piece = piece + 1 == npieces ? 0 : piece + 1;
    };
    if (w == 4) {
      issue(0); issue(1);
      for (int it = 0; it < iters; ++it) {
        if (it + 2 < iters) { issue(it + 2); asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); }   // stage `it` landed
        else if (it + 1 < iters) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const uint32_t a0 = lds0 + (uint32_t)((it & 3) * 32768 + (w & 1) * 16384 + lane * 16);
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a0), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 4) acc ^= v[i].x;
      }
    }
  } else if constexpr (MODE == 6 || MODE == 7) {
    // split path: half of every 32-KiB stage by LDS-DMA (the LDS operand), half straight into registers (an operand stored
    // in fragment order needs no LDS); 8 vector-memory operations per wave per stage as before, 3 stages in flight
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lp_smem;
    const uint32_t voff = lane * 16;
    uint4 rb[4];
    auto issue = [&](int it) {
      const unsigned char* b = base + (size_t)piece * 32768u + w * 8192;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t dst = lds0 + (uint32_t)((it & 3) * 16384 + (w * 4 + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(b + i * 1024), "s"(dst) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[i]) : "v"(voff), "s"(b + 4096 + i * 1024) : "memory");
      piece = piece + 1 == npieces ? 0 : piece + 1;
    };
    issue(0); issue(1); issue(2);
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __syncthreads();
      if (it + 3 < iters) issue(it + 3);
      if constexpr (MODE == 6) {
        const uint32_t a0 = lds0 + (uint32_t)((it & 3) * 16384 + lane * 16);
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a0), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 4) acc ^= v[i].x;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= rb[0].x;
  } else {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lp_smem;
    const uint32_t voff = lane * 16;
    auto issue = [&](int it) {
      const unsigned char* b = base + (size_t)piece * 32768u + w * 8192;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t dst = lds0 + (uint32_t)((it & 3) * 32768 + (w * 8 + i) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(b + i * 1024), "s"(dst) : "memory");
      }
      piece = piece + 1 == npieces ? 0 : piece + 1;
    };
    if constexpr (MODE != 3 && MODE != 9) { issue(0); issue(1); issue(2); }
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // stage `it` landed (two later stages may be in flight)
      __syncthreads();
      if constexpr (MODE != 3 && MODE != 9) { if (it + 3 < iters) issue(it + 3); }
      if constexpr (MODE == 9) {
        // the GEMM's fragment pattern: 32 rows of 128 B, chunk (2s+hi) ^ ((row>>1)&7); 4 sub-steps x (2 A + 2 B) blocks
        const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
        uint4 v[16];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const uint32_t o = lds0 + (uint32_t)((it & 3) * 32768 + (((2 * sub + hi) ^ sw) << 4) + l31 * 128);
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[sub * 4 + 0]) : "v"(o), "n"(0));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[sub * 4 + 1]) : "v"(o), "n"(4096));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[sub * 4 + 2]) : "v"(o), "n"(16384));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[sub * 4 + 3]) : "v"(o), "n"(16384 + 4096));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 4) acc ^= v[i].x;
      } else if constexpr (MODE >= 2) {
        const uint32_t a0 = lds0 + (uint32_t)((it & 3) * 32768 + (w & 1) * 16384 + lane * 16);
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a0), "n"(i * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 4) acc ^= v[i].x;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
}  // namespace

extern "C" int dqnhip_test_loadpath(int32_t mode, int32_t blocks, int32_t region_kb, int32_t iters, int32_t launches,
                                    float* avg_us, float* tb_per_s) {
  if (mode < 0 || mode > 9 || blocks < 8 || region_kb < 32 || (region_kb & 31) || iters < 4 || launches < 1) return 1;
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t region = (size_t)region_kb * 1024;
  unsigned char* src; uint32_t* sink;
  CK(hipMalloc(&src, region * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemsetAsync(src, 1, region * 8, s));
  auto launch = [&]() {
    const int lds = mode == 0 ? 0 : 4 * 32768;
    if (mode == 0) hipLaunchKernelGGL(k_loadpath<0>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 1) hipLaunchKernelGGL(k_loadpath<1>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 2) hipLaunchKernelGGL(k_loadpath<2>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 3) hipLaunchKernelGGL(k_loadpath<3>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 4) hipLaunchKernelGGL(k_loadpath<4>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 5) hipLaunchKernelGGL(k_loadpath<5>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 6) hipLaunchKernelGGL(k_loadpath<6>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 7) hipLaunchKernelGGL(k_loadpath<7>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
    else if (mode == 8) hipLaunchKernelGGL(k_loadpath<8>, dim3(blocks), dim3(320), lds, s, src, (uint32_t)region, iters, sink);
    else hipLaunchKernelGGL(k_loadpath<9>, dim3(blocks), dim3(256), lds, s, src, (uint32_t)region, iters, sink);
  };
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loadpath<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
  for (int i = 0; i < launches; ++i) launch();
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  const float us = ms * 1000.f / launches;
  if (avg_us) *avg_us = us;
  if (tb_per_s) *tb_per_s = (float)((double)blocks * iters * 32768.0 / (us * 1e-6) / 1e12);
  hipFree(src); hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  return 0;
}


// ---- launch-floor probe: what one link of a dependent kernel chain costs inside a replayed hipGraph, by launch shape ----
// variant bits: 1 = 640-byte kernarg (a GemmBatch by value) instead of one pointer; 2 = dynamic LDS (lds_bytes) ; 4 = a body with
// one dependent global round trip (load -> store); 8 = 1024 threads per block.  blocks = grid size.
namespace {
struct FloorArgs { float* p; int pad[158]; };
__global__ void k_floor_small(float* p, int dep) {
  if (dep) { const float v = p[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFF]; if (v == 123.456f) p[0] = v; }
}
__global__ void k_floor_big(FloorArgs a, int dep) {
  if (dep) { const float v = a.p[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFF]; if (v == 123.456f) a.p[0] = v + a.pad[7]; }
}
__global__ void k_floor_lds(float* p, int dep) {
  extern __shared__ float fl_sm[];
  if (dep) { const float v = p[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFF]; fl_sm[threadIdx.x] = v; if (v == 123.456f) p[0] = fl_sm[(threadIdx.x + 1) & 255]; }
}
__global__ void k_floor_big_lds(FloorArgs a, int dep) {
  extern __shared__ float fl_sm[];
  if (dep) { const float v = a.p[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFF]; fl_sm[threadIdx.x] = v; if (v == 123.456f) a.p[0] = fl_sm[(threadIdx.x + 1) & 255] + a.pad[7]; }
}
}  // namespace
extern "C" int dqnhip_test_launch_floor(int32_t variant, int32_t blocks, int32_t lds_bytes, int32_t chain, int32_t iters, float* us_per_kernel) {
  if (blocks < 1 || chain < 1 || iters < 1 || lds_bytes < 0 || lds_bytes > 160 * 1024) return 1;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* buf; CK(hipMalloc(&buf, 65536 * 4)); CK(hipMemsetAsync(buf, 0, 65536 * 4, s));
  const bool big = variant & 1, lds = variant & 2; const int dep = (variant & 4) ? 1 : 0; const int nt = (variant & 8) ? 1024 : 256;
  if (lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_floor_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_floor_big_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  }
  FloorArgs fa{}; fa.p = buf;
  auto one = [&]() {
    if (big && lds) hipLaunchKernelGGL(k_floor_big_lds, dim3(blocks), dim3(nt), lds_bytes, s, fa, dep);
    else if (big) hipLaunchKernelGGL(k_floor_big, dim3(blocks), dim3(nt), 0, s, fa, dep);
    else if (lds) hipLaunchKernelGGL(k_floor_lds, dim3(blocks), dim3(nt), lds_bytes, s, buf, dep);
    else hipLaunchKernelGGL(k_floor_small, dim3(blocks), dim3(nt), 0, s, buf, dep);
  };
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < chain; ++i) one();
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphDestroy(graph));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(exec, s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, s));
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  *us_per_kernel = ms * 1e3f / ((float)iters * chain);
  hipGraphExecDestroy(exec); hipEventDestroy(e0); hipEventDestroy(e1); hipFree(buf); hipStreamDestroy(s);
  return 0;
}
