// gemm_direct.hip.h — "direct-to-register, split-K-across-waves" fp32 MFMA GEMMs
// for the small-M (minibatch 32..4096) layers of the actor/critic towers.
//
// Why a second family: at M = 256 a 1024x1024 layer is only 0.54 GFLOP.  An
// LDS-staged tile kernel (gemm_mfma.hip.h) needs one workgroup barrier per K
// tile and can only offer 128-256 workgroups; measured 16-25 us per layer
// (profiles/r01_v1_*).  Here instead:
//   * every output tile is computed by ONE workgroup of 4 waves, each wave
//     accumulating the WHOLE tile over its own quarter of the reduction
//     range (in-workgroup split-K).  The waves never synchronise in the main
//     loop — four independent load->MFMA pipelines per CU, one per SIMD;
//   * operands go global/L2 -> VGPR directly in MFMA fragment layout with
//     16-byte loads, no LDS round trip: a "k-contiguous" operand (KC) is read
//     as float4 along k (4 consecutive k-steps of one row per lane); a
//     "k-strided" operand (KS) is read as float4 along the free dimension
//     (whole 256-B rows per 16 lanes) and its four components feed four
//     interleaved MFMAs (output columns 4i+c);
//   * a 4-deep register ring keeps >= 2048 MFMA-cycles of loads in flight;
//   * one LDS pass at the end adds the four waves' partial tiles in fixed order
//     (w0+w1)+(w2+w3) -> deterministic, no atomics; the epilogue (bias +
//     leaky ReLU / ReLU' mask / bias-gradient + sum-of-squares partial) is
//     applied by the wave that reduces the accumulator.
// The smallest tiles (32x32 fwd, 64x16 dgrad, 64x64 wgrad) give 256
// workgroups for ONE 256x1024x1024 layer, i.e. a single layer fills the chip.
//
// Same GemmProblem/GemmBatch interface and the same three modes as
// gemm_mfma.hip.h (C[q][p] = sum_k P(p,k) Q(q,k), p contiguous).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "gemm_common.hip.h"

namespace dqnhip {

// ---- shared pieces ----------------------------------------------------------------

__device__ __forceinline__ void tile_of_problem(const GemmProblem& pr, int b, int& tile_p, int& tile_q) {
  if ((pr.tiles_p & 7) == 0) {       // same P panel (weight slice) -> same XCD L2 (b % 8)
    const int xcd = b & 7, j = b >> 3;
    tile_q = j % pr.tiles_q;
    tile_p = (j / pr.tiles_q) * 8 + xcd;
  } else {
    tile_q = b % pr.tiles_q;
    tile_p = b / pr.tiles_q;
  }
}
__device__ __forceinline__ void tile_of_block(const GemmBatch& batch, int& pi, int& tile_p, int& tile_q) {
  int b = blockIdx.x;
  pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < batch.n && b >= batch.prob[i].tile_base) pi = i;
  const GemmProblem& pr = batch.prob[pi];
  tile_of_problem(pr, b - pr.tile_base, tile_p, tile_q);
}

// Each wave parks its NACC accumulators in LDS (lane-linear: conflict free), then wave w
// returns, for every accumulator e with (e & 3) == w, the fixed-order sum over the 4 waves.
template <int NACC>
__device__ __forceinline__ void park_accumulators(float* smem, const f32x4 (&acc)[NACC], int wave, int lane) {
  f32x4* s = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int e = 0; e < NACC; ++e) s[(wave * NACC + e) * 64 + lane] = acc[e];
}
template <int NACC>
__device__ __forceinline__ f32x4 reduce_accumulator(const float* smem, int e, int lane) {
  const f32x4* s = reinterpret_cast<const f32x4*>(smem);
  const f32x4 a0 = s[(0 * NACC + e) * 64 + lane], a1 = s[(1 * NACC + e) * 64 + lane];
  const f32x4 a2 = s[(2 * NACC + e) * 64 + lane], a3 = s[(3 * NACC + e) * 64 + lane];
  f32x4 r;
  r.x = (a0.x + a1.x) + (a2.x + a3.x); r.y = (a0.y + a1.y) + (a2.y + a3.y);
  r.z = (a0.z + a1.z) + (a2.z + a3.z); r.w = (a0.w + a1.w) + (a2.w + a3.w);
  return r;
}

#define DQN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// hipcc otherwise sinks every prefetch load below the whole MFMA block of an iteration
// (collapsing the register ring: measured); pin the COMPUTE/LOAD interleave.
#define DQN_PIN() __builtin_amdgcn_sched_barrier(0)

// GemmProblem::seed_w: the tower-top gradient of the dq = -1 pass, from the finished activations v of this lane
// (k_head_bwd<1>'s arithmetic: s0 = fma(-1, w, 0) = -w, then * lrelu'(x))
__device__ __forceinline__ void store_head_seed(const GemmProblem& pr, int q, int p, const f32x4& v, const f32x4& sw) {
  f32x4 dz;
  dz.x = (-sw.x) * lrelu_mask(v.x); dz.y = (-sw.y) * lrelu_mask(v.y);
  dz.z = (-sw.z) * lrelu_mask(v.z); dz.w = (-sw.w) * lrelu_mask(v.w);
  *reinterpret_cast<f32x4*>(pr.C2 + (size_t)q * pr.ldc + p) = dz;
}

// GemmProblem::dot_w: this lane's four finished activations v against the head weights dw, summed over the four lane groups
// (16 columns), written by lane group 0
__device__ __forceinline__ void store_head_dot(const GemmProblem& pr, int q, int p, int lg, const f32x4& v, const f32x4& dw) {
  float d = fmaf(v.x, dw.x, 0.0f); d = fmaf(v.y, dw.y, d); d = fmaf(v.z, dw.z, d); d = fmaf(v.w, dw.w, d);
  d += __shfl_xor(d, 16, 64);
  d += __shfl_xor(d, 32, 64);
  if (lg == 0) pr.dot_out[(size_t)q * (pr.Pdim >> 4) + (p >> 4)] = d;
}

// ================================ FWD ================================================
// Y[m][n] = lrelu(sum_k X[m][k] W[n][k] + b[n]).  P = W (KC, 16-row blocks), Q = X (KC).
// Tile = (16*TP) x (16*TQ).  Kred % 64 == 0.
template <int TP, int TQ>
__device__ __forceinline__ void fwd_direct_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NACC = TP * TQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 16 * TP, q0 = tile_q * 16 * TQ;
  const int Kw = pr.Kred >> 2;              // this wave's share of the reduction
  const int nkb = Kw >> 4;                  // 16-wide k blocks
  const float* pp[TP];
  const float* qp[TQ];
#pragma unroll
  for (int c = 0; c < TP; ++c) pp[c] = pr.P + (size_t)(p0 + c * 16 + li) * pr.ldp + wave * Kw + lg * 4;
#pragma unroll
  for (int a = 0; a < TQ; ++a) qp[a] = pr.Q + (size_t)(q0 + a * 16 + li) * pr.ldq + wave * Kw + lg * 4;

  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 rp[4][TP], rq[4][TQ];

#define FWD_LOAD(slot, kb)                                                              \
  {                                                                                     \
    _Pragma("unroll") for (int c = 0; c < TP; ++c)                                      \
        rp[slot][c] = *reinterpret_cast<const f32x4*>(pp[c] + ((kb) << 4));            \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
        rq[slot][a] = *reinterpret_cast<const f32x4*>(qp[a] + ((kb) << 4));            \
  }
#define FWD_COMPUTE(slot)                                                               \
  {                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
    _Pragma("unroll") for (int c = 0; c < TP; ++c)                                      \
        acc[a * TP + c] = DQN_MFMA(rp[slot][c][s], rq[slot][a][s], acc[a * TP + c]);    \
  }

  const int nkb4 = nkb & ~3;
  if (nkb4 > 0) {
    FWD_LOAD(0, 0) FWD_LOAD(1, 1) FWD_LOAD(2, 2) FWD_LOAD(3, 3)
    int kb = 0;
    for (; kb + 4 < nkb4; kb += 4) {
      FWD_COMPUTE(0) DQN_PIN(); FWD_LOAD(0, kb + 4) DQN_PIN();
      FWD_COMPUTE(1) DQN_PIN(); FWD_LOAD(1, kb + 5) DQN_PIN();
      FWD_COMPUTE(2) DQN_PIN(); FWD_LOAD(2, kb + 6) DQN_PIN();
      FWD_COMPUTE(3) DQN_PIN(); FWD_LOAD(3, kb + 7) DQN_PIN();
    }
    FWD_COMPUTE(0) FWD_COMPUTE(1) FWD_COMPUTE(2) FWD_COMPUTE(3)
  }
  // the (up to three) remaining steps: every load first, then the MFMAs (the first tower layer has K_in = 64 / 128,
  // i.e. ONLY these steps: one load round trip instead of one per step)
  {
    const int rem = nkb - nkb4;
    if (rem > 0) FWD_LOAD(0, nkb4)
    if (rem > 1) FWD_LOAD(1, nkb4 + 1)
    if (rem > 2) FWD_LOAD(2, nkb4 + 2)
    if (rem > 0) FWD_COMPUTE(0)
    if (rem > 1) FWD_COMPUTE(1)
    if (rem > 2) FWD_COMPUTE(2)
  }
#undef FWD_LOAD
#undef FWD_COMPUTE

  // this wave's bias (and head-seed) pieces, requested ahead of the cross-wave reduction
  constexpr int NBV = (NACC + 3) / 4;
  f32x4 bvp[NBV], swp[NBV], dwp[NBV];
  if (pr.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) bvp[j] = *reinterpret_cast<const f32x4*>(pr.bias + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  if (pr.seed_w != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) swp[j] = *reinterpret_cast<const f32x4*>(pr.seed_w + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  if (pr.dot_w != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) dwp[j] = *reinterpret_cast<const f32x4*>(pr.dot_w + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  if (pr.xcopy_dst != nullptr && (int)threadIdx.x < 16 * TP) {      // (column j by the workgroup of row tile j mod tiles_q: one or two dwords per thread)
    const float* src = pr.P + (size_t)(p0 + threadIdx.x) * pr.ldp + pr.xcopy_col;
    for (int j = tile_q; j < pr.xcopy_n; j += pr.tiles_q) pr.xcopy_dst[(size_t)j * pr.Pdim + p0 + threadIdx.x] = src[j];
  }
  park_accumulators<NACC>(smem, acc, wave, lane);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NACC; ++e) {
    if ((e & 3) == wave) {
      const int a = e / TP, c = e % TP;
      f32x4 v = reduce_accumulator<NACC>(smem, e, lane);
      const int q = q0 + a * 16 + li, p = p0 + c * 16 + (lg << 2);
      if (pr.bias != nullptr) {
        const f32x4 bv = bvp[e >> 2];
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (pr.relu) { v.x = lrelu_fwd(v.x); v.y = lrelu_fwd(v.y); v.z = lrelu_fwd(v.z); v.w = lrelu_fwd(v.w); }
      *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
      if (pr.seed_w != nullptr) store_head_seed(pr, q, p, v, swp[e >> 2]);
      if (pr.dot_w != nullptr) store_head_dot(pr, q, p, lg, v, dwp[e >> 2]);
    }
  }
}

// ================================ DGRAD ==============================================
// dX[m][j] = (sum_n dY[m][n] W[n][j]) * lrelu'(act[m][j]).  P = W (KS, 64-wide blocks of j),
// Q = dY (KC, 16-row blocks of m).  Tile = (64*TPB) x (16*TQ).  Kred (= n) % 64 == 0.
template <int TPB, int TQ>
__device__ __forceinline__ void dgrad_direct_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NACC = TPB * 4 * TQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 64 * TPB, q0 = tile_q * 16 * TQ;
  const int Kw = pr.Kred >> 2;
  const int nkb = Kw >> 4;
  // P rows are reduction indices: lane group lg owns n = base + kb*16 + 4*lg + s at step s
  const float* pp = pr.P + (size_t)(wave * Kw + lg * 4) * pr.ldp + p0 + li * 4;
  const float* qp[TQ];
#pragma unroll
  for (int a = 0; a < TQ; ++a) qp[a] = pr.Q + (size_t)(q0 + a * 16 + li) * pr.ldq + wave * Kw + lg * 4;
  const size_t ldp = pr.ldp;

  f32x4 acc[NACC];   // index ((a*TPB + b)*4 + pc)
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 rp[4][4][TPB], rq[4][TQ];

#define DG_LOAD(slot, kb)                                                               \
  {                                                                                     \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
        rq[slot][a] = *reinterpret_cast<const f32x4*>(qp[a] + ((kb) << 4));            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
        rp[slot][s][b] = *reinterpret_cast<const f32x4*>(pp + (size_t)(((kb) << 4) + s) * ldp + b * 64); \
  }
#define DG_COMPUTE(slot)                                                                \
  {                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                                    \
        acc[(a * TPB + b) * 4 + pc] =                                                   \
            DQN_MFMA(rp[slot][s][b][pc], rq[slot][a][s], acc[(a * TPB + b) * 4 + pc]);  \
  }

  const int nkb4 = nkb & ~3;
  if (nkb4 > 0) {
    DG_LOAD(0, 0) DG_LOAD(1, 1) DG_LOAD(2, 2) DG_LOAD(3, 3)
    int kb = 0;
    for (; kb + 4 < nkb4; kb += 4) {
      DG_COMPUTE(0) DQN_PIN(); DG_LOAD(0, kb + 4) DQN_PIN();
      DG_COMPUTE(1) DQN_PIN(); DG_LOAD(1, kb + 5) DQN_PIN();
      DG_COMPUTE(2) DQN_PIN(); DG_LOAD(2, kb + 6) DQN_PIN();
      DG_COMPUTE(3) DQN_PIN(); DG_LOAD(3, kb + 7) DQN_PIN();
    }
    DG_COMPUTE(0) DG_COMPUTE(1) DG_COMPUTE(2) DG_COMPUTE(3)
  }
  for (int kb = nkb4; kb < nkb; ++kb) { DG_LOAD(0, kb) DG_COMPUTE(0) }
#undef DG_LOAD
#undef DG_COMPUTE

  // this wave's ReLU' mask pieces, requested ahead of the cross-wave reduction
  constexpr int NMK = (TQ * TPB + 3) / 4;
  f32x4 mk[NMK][4];
  if (pr.mask != nullptr) {
#pragma unroll
    for (int j = 0; j < NMK; ++j) {
      const int ab = j * 4 + wave;
      if (ab < TQ * TPB) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          mk[j][r] = *reinterpret_cast<const f32x4*>(pr.mask + (size_t)(q0 + (ab / TPB) * 16 + li) * pr.ldm + p0 + (ab % TPB) * 64 + (lg << 4) + (r << 2));
      }
    }
  }
  park_accumulators<NACC>(smem, acc, wave, lane);
  __syncthreads();
  // accumulators (a,b,pc=0..3) form float4s over pc: reduce them as a group of 4
#pragma unroll
  for (int ab = 0; ab < TQ * TPB; ++ab) {
    if ((ab & 3) == wave) {
      const int a = ab / TPB, b = ab % TPB;
      f32x4 r0 = reduce_accumulator<NACC>(smem, ab * 4 + 0, lane);
      f32x4 r1 = reduce_accumulator<NACC>(smem, ab * 4 + 1, lane);
      f32x4 r2 = reduce_accumulator<NACC>(smem, ab * 4 + 2, lane);
      f32x4 r3 = reduce_accumulator<NACC>(smem, ab * 4 + 3, lane);
      const int q = q0 + a * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = p0 + b * 64 + (lg << 4) + (r << 2);
        f32x4 v = f32x4{r0[r], r1[r], r2[r], r3[r]};
        if (pr.mask != nullptr) {
          const f32x4 mv = mk[ab >> 2][r];
          v.x *= lrelu_mask(mv.x); v.y *= lrelu_mask(mv.y); v.z *= lrelu_mask(mv.z); v.w *= lrelu_mask(mv.w);
        }
        *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
      }
    }
  }
}

// ---- DGRAD, half-width tiles: 32 input columns x 16 rows (round 6 experiment) ---------------------------
// Twice the workgroups of dgrad_direct_body<1, 1> (512 for a 256 x 1024 x 1024 layer), each with half the MFMA chain and a quarter
// of the LDS: two (or more) co-resident per CU.  P = W read as 8-byte pieces (two columns per lane), Q = dY as in dgrad_direct_body.
// Harness variant 8 of dqnhip_test_gemm mode 1; measured against variants 1 (64 x 16 direct) and 5 (64 x 16, dY through the LDS
// transpose — the learner's): profiles/r06_dgrad_half_tiles.txt.
__device__ __forceinline__ void dgrad_direct32_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NACC = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 32, q0 = tile_q * 16;
  const int Kw = pr.Kred >> 2;
  const int nkb = Kw >> 4;
  const float* pp = pr.P + (size_t)(wave * Kw + lg * 4) * pr.ldp + p0 + li * 2;
  const float* qp = pr.Q + (size_t)(q0 + li) * pr.ldq + wave * Kw + lg * 4;
  const size_t ldp = pr.ldp;
  f32x4 acc[NACC];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
  f32x2 rp[4][4]; f32x4 rq[4];
#define D32_LOAD(slot, kb)                                                              \
  {                                                                                     \
    rq[slot] = *reinterpret_cast<const f32x4*>(qp + ((kb) << 4));                      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
        rp[slot][s] = *reinterpret_cast<const f32x2*>(pp + (size_t)(((kb) << 4) + s) * ldp); \
  }
#define D32_COMPUTE(slot)                                                               \
  {                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
    _Pragma("unroll") for (int pc = 0; pc < 2; ++pc)                                    \
        acc[pc] = DQN_MFMA(rp[slot][s][pc], rq[slot][s], acc[pc]);                      \
  }
  const int nkb4 = nkb & ~3;
  if (nkb4 > 0) {
    D32_LOAD(0, 0) D32_LOAD(1, 1) D32_LOAD(2, 2) D32_LOAD(3, 3)
    int kb = 0;
    for (; kb + 4 < nkb4; kb += 4) {
      D32_COMPUTE(0) DQN_PIN(); D32_LOAD(0, kb + 4) DQN_PIN();
      D32_COMPUTE(1) DQN_PIN(); D32_LOAD(1, kb + 5) DQN_PIN();
      D32_COMPUTE(2) DQN_PIN(); D32_LOAD(2, kb + 6) DQN_PIN();
      D32_COMPUTE(3) DQN_PIN(); D32_LOAD(3, kb + 7) DQN_PIN();
    }
    D32_COMPUTE(0) D32_COMPUTE(1) D32_COMPUTE(2) D32_COMPUTE(3)
  }
  for (int kb = nkb4; kb < nkb; ++kb) { D32_LOAD(0, kb) D32_COMPUTE(0) }
#undef D32_LOAD
#undef D32_COMPUTE
  // C/D map of acc[pc]: lane (li, lg), register r = dX[row q0 + li][column p0 + 2 (4 lg + r) + pc]: 8 consecutive columns per lane
  f32x4 m0 = f32x4{1.f, 1.f, 1.f, 1.f}, m1 = m0;
  if (wave == 0 && pr.mask != nullptr) {
    const float* mp = pr.mask + (size_t)(q0 + li) * pr.ldm + p0 + (lg << 3);
    m0 = *reinterpret_cast<const f32x4*>(mp); m1 = *reinterpret_cast<const f32x4*>(mp + 4);
  }
  park_accumulators<NACC>(smem, acc, wave, lane);
  __syncthreads();
  if (wave == 0) {
    const f32x4 a0 = reduce_accumulator<NACC>(smem, 0, lane), a1 = reduce_accumulator<NACC>(smem, 1, lane);
    f32x4 v0 = f32x4{a0.x, a1.x, a0.y, a1.y}, v1 = f32x4{a0.z, a1.z, a0.w, a1.w};
    if (pr.mask != nullptr) {
      v0.x *= lrelu_mask(m0.x); v0.y *= lrelu_mask(m0.y); v0.z *= lrelu_mask(m0.z); v0.w *= lrelu_mask(m0.w);
      v1.x *= lrelu_mask(m1.x); v1.y *= lrelu_mask(m1.y); v1.z *= lrelu_mask(m1.z); v1.w *= lrelu_mask(m1.w);
    }
    float* c = pr.C + (size_t)(q0 + li) * pr.ldc + p0 + (lg << 3);
    *reinterpret_cast<f32x4*>(c) = v0; *reinterpret_cast<f32x4*>(c + 4) = v1;
  }
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void gemm_dgrad_direct32(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  dgrad_direct32_body(batch.prob[pi], tile_p, tile_q, smem);
}

// ---- DGRAD, narrow: 16 input columns per workgroup ------------------------------------------
// The critic's first-layer input gradient is consumed only in the 10 action columns (inverting
// gradients, src/dqn.cpp:924-957): instead of 64-wide tiles over the whole 128-column panel (32
// workgroups, 4.2 us of MFMA per wave) only the 16-column tiles that contain those columns are
// computed, a quarter of the MFMA chain per wave.  P = W (one column per lane, scalar loads),
// Q = dY (16 rows).  Same K split over the 4 waves, same fixed-order reduction.
// dgrad_narrow_tile: the reduced (and masked) 16 x 16 tile in wave 0's lanes — lane (li, lg), register r = dX[row q0 + li][column
// p0 + 4 lg + r]; the other waves return zeros.  dgrad_narrow_body stores it.
// NS: register ring depth in 16-k steps (a wave's quarter of a 1024-deep reduction is 16 steps: NS = 8 -> two load round trips
// instead of four; the order in which the MFMAs accumulate does not depend on it)
template <int NS = 4>
__device__ __forceinline__ f32x4 dgrad_narrow_tile(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NACC = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 16, q0 = tile_q * 16;
  const int Kw = pr.Kred >> 2;
  const int nkb = Kw >> 4;
  const float* pp = pr.P + (size_t)(wave * Kw + lg * 4) * pr.ldp + p0 + li;
  const float* qp = pr.Q + (size_t)(q0 + li) * pr.ldq + wave * Kw + lg * 4;
  const size_t ldp = pr.ldp;
  f32x4 acc[NACC];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rp[NS][4]; f32x4 rq[NS];
#define DN_LOAD(slot, kb)                                                               \
  {                                                                                     \
    rq[slot] = *reinterpret_cast<const f32x4*>(qp + ((kb) << 4));                      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) rp[slot][s] = pp[(size_t)(((kb) << 4) + s) * ldp]; \
  }
#define DN_COMPUTE(slot)                                                                \
  { _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[0] = DQN_MFMA(rp[slot][s], rq[slot][s], acc[0]); }
  const int nkbN = nkb - nkb % NS;
  if (nkbN > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { DN_LOAD(i, i) DQN_PIN(); }
    int kb = 0;
    for (; kb + NS < nkbN; kb += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { DN_COMPUTE(i) DQN_PIN(); DN_LOAD(i, kb + NS + i) DQN_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { DN_COMPUTE(i) }
  }
  for (int kb = nkbN; kb < nkb; ++kb) { DN_LOAD(0, kb) DN_COMPUTE(0) }
#undef DN_LOAD
#undef DN_COMPUTE
  f32x4 mv = f32x4{0.f, 0.f, 0.f, 0.f};          // requested ahead of the cross-wave reduction
  if (wave == 0 && pr.mask != nullptr) mv = *reinterpret_cast<const f32x4*>(pr.mask + (size_t)(q0 + li) * pr.ldm + p0 + (lg << 2));
  park_accumulators<NACC>(smem, acc, wave, lane);
  __syncthreads();
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    // C/D map: lane (li, lg) register r = C[i = 4 lg + r][j = li] = dX[row q0 + li][column p0 + 4 lg + r]
    v = reduce_accumulator<NACC>(smem, 0, lane);
    if (pr.mask != nullptr) {
      v.x *= lrelu_mask(mv.x); v.y *= lrelu_mask(mv.y); v.z *= lrelu_mask(mv.z); v.w *= lrelu_mask(mv.w);
    }
  }
  return v;
}
// The same tile from the fp16 learner's operands (round 6): P = the fp16 weight mirror W16[n][k_in] (one column per lane), Q = the
// scaled fp16 gradient panel dY16[rows][n].  fp16 x fp16 products are exact in fp32, so this is the fp16-MFMA dgrad's arithmetic up
// to the order of the fp32 additions; the caller removes the loss scale.  No mask (the layer's input has no ReLU).
struct NarrowTile16 { const _Float16* P; int ldp; const _Float16* Q; int ldq; int Kred; };
template <int NS = 4>
__device__ __forceinline__ f32x4 dgrad_narrow_tile16(const NarrowTile16& pr, int tile_q, float* smem) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int q0 = tile_q * 16;
  const int Kw = pr.Kred >> 2;
  const int nkb = Kw >> 4;
  const _Float16* pp = pr.P + (size_t)(wave * Kw + lg * 4) * pr.ldp + li;
  const _Float16* qp = pr.Q + (size_t)(q0 + li) * pr.ldq + wave * Kw + lg * 4;
  const size_t ldp = pr.ldp;
  f32x4 acc[1];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  _Float16 rp[NS][4]; h4 rq[NS];
#define DN_LOAD(slot, kb)                                                               \
  {                                                                                     \
    rq[slot] = *reinterpret_cast<const h4*>(qp + ((kb) << 4));                         \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) rp[slot][s] = pp[(size_t)(((kb) << 4) + s) * ldp]; \
  }
#define DN_COMPUTE(slot)                                                                \
  { _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[0] = DQN_MFMA((float)rp[slot][s], (float)rq[slot][s], acc[0]); }
  const int nkbN = nkb - nkb % NS;
  if (nkbN > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { DN_LOAD(i, i) DQN_PIN(); }
    int kb = 0;
    for (; kb + NS < nkbN; kb += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { DN_COMPUTE(i) DQN_PIN(); DN_LOAD(i, kb + NS + i) DQN_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { DN_COMPUTE(i) }
  }
  for (int kb = nkbN; kb < nkb; ++kb) { DN_LOAD(0, kb) DN_COMPUTE(0) }
#undef DN_LOAD
#undef DN_COMPUTE
  park_accumulators<1>(smem, acc, wave, lane);
  __syncthreads();
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave == 0) v = reduce_accumulator<1>(smem, 0, lane);
  return v;
}
__device__ __forceinline__ void dgrad_narrow_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  const f32x4 v = dgrad_narrow_tile<4>(pr, tile_p, tile_q, smem);
  if ((threadIdx.x >> 6) == 0) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int q = tile_q * 16 + li, p = tile_p * 16 + (lg << 2);
    *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
  }
}

// ================================ WGRAD ==============================================
// dW[n][j] = sum_m dY[m][n] X[m][j];  db[n] = sum_m dY[m][n].  P = X (KS, 64-wide blocks of
// j), Q = dY (KS, 64-wide blocks of n).  Tile = (64*TPB) x (64*TQB).  Kred (= rows m) % 16 == 0.
constexpr int kWgradRing = 4;   // register ring of wgrad_direct_body (8 measured slower inside the pair kernel: 16.4 vs 15.3 us)
template <int TPB, int TQB>
__device__ __forceinline__ void wgrad_direct_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NACC = TPB * 4 * TQB * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 64 * TPB, q0 = tile_q * 64 * TQB;
  const int Kw = pr.Kred >> 2;
  const int nst = Kw >> 2;                 // steps of 4 rows
  const float* pp = pr.P + (size_t)(wave * Kw + lg) * pr.ldp + p0 + li * 4;
  const float* qp = pr.Q + (size_t)(wave * Kw + lg) * pr.ldq + q0 + li * 4;
  const size_t ldp = pr.ldp, ldq = pr.ldq;
  const bool want_db = (pr.db != nullptr) && (tile_p == 0);

  f32x4 acc[NACC];   // index (((d*4 + qc)*TPB + b)*4 + pc)
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbacc[TQB];
#pragma unroll
  for (int d = 0; d < TQB; ++d) dbacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  // register ring of NS steps (4 rows of X and dY each): NS-1 steps of lookahead
  constexpr int NS = kWgradRing;
  f32x4 rp[NS][TPB], rq[NS][TQB];

#define WG_LOAD(slot, st)                                                               \
  {                                                                                     \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
        rp[slot][b] = *reinterpret_cast<const f32x4*>(pp + (size_t)((st) << 2) * ldp + b * 64); \
    _Pragma("unroll") for (int d = 0; d < TQB; ++d)                                     \
        rq[slot][d] = *reinterpret_cast<const f32x4*>(qp + (size_t)((st) << 2) * ldq + d * 64); \
  }
#define WG_COMPUTE(slot)                                                                \
  {                                                                                     \
    _Pragma("unroll") for (int d = 0; d < TQB; ++d) {                                   \
      dbacc[d].x += rq[slot][d].x; dbacc[d].y += rq[slot][d].y;                         \
      dbacc[d].z += rq[slot][d].z; dbacc[d].w += rq[slot][d].w;                         \
      _Pragma("unroll") for (int qc = 0; qc < 4; ++qc)                                  \
      _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                   \
      _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                                  \
          acc[((d * 4 + qc) * TPB + b) * 4 + pc] = DQN_MFMA(                            \
              rp[slot][b][pc], rq[slot][d][qc], acc[((d * 4 + qc) * TPB + b) * 4 + pc]); \
    }                                                                                   \
  }

  const int nstN = nst - nst % NS;
  if (nstN > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { WG_LOAD(i, i) DQN_PIN(); }
    int st = 0;
    for (; st + NS < nstN; st += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { WG_COMPUTE(i) DQN_PIN(); WG_LOAD(i, st + NS + i) DQN_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { WG_COMPUTE(i) }
  }
  {   // the (up to NS - 1) remaining steps: every load first (minibatch 32: these two steps are the whole reduction)
    const int rem = nst - nstN;
    if (rem > 0) WG_LOAD(0, nstN)
    if (rem > 1) WG_LOAD(1, nstN + 1)
    if (rem > 2) WG_LOAD(2, nstN + 2)
    if (rem > 0) WG_COMPUTE(0)
    if (rem > 1) WG_COMPUTE(1)
    if (rem > 2) WG_COMPUTE(2)
  }
#undef WG_LOAD
#undef WG_COMPUTE

  park_accumulators<NACC>(smem, acc, wave, lane);
  float* sdb = smem + 4 * NACC * 64 * 4;      // [4 waves][TQB][16 li] float4
  if (want_db) {
#pragma unroll
    for (int d = 0; d < TQB; ++d) {
      f32x4 v = dbacc[d];
      // add the 4 lane groups (rows m+0..3): lanes l, l^16, l^32, l^48
      v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
      v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
      if (lg == 0) reinterpret_cast<f32x4*>(sdb)[(wave * TQB + d) * 16 + li] = v;
    }
  }
  __syncthreads();
  float ssq = 0.0f;
  // accumulators (d,qc,b,pc=0..3) form float4s over pc
#pragma unroll
  for (int g4 = 0; g4 < TQB * 4 * TPB; ++g4) {
    if ((g4 & 3) == wave) {
      const int b = g4 % TPB, dq = g4 / TPB, qc = dq & 3, d = dq >> 2;
      f32x4 r0 = reduce_accumulator<NACC>(smem, g4 * 4 + 0, lane);
      f32x4 r1 = reduce_accumulator<NACC>(smem, g4 * 4 + 1, lane);
      f32x4 r2 = reduce_accumulator<NACC>(smem, g4 * 4 + 2, lane);
      f32x4 r3 = reduce_accumulator<NACC>(smem, g4 * 4 + 3, lane);
      const int n = q0 + d * 64 + (li << 2) + qc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = p0 + b * 64 + (lg << 4) + (r << 2);
        const f32x4 v = f32x4{r0[r], r1[r], r2[r], r3[r]};
        ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq); ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
        *reinterpret_cast<f32x4*>(pr.C + (size_t)n * pr.ldc + p) = v;
      }
    }
  }
  if (want_db && wave == 0 && lane < 16 * TQB) {
    const int d = lane >> 4, l16 = lane & 15;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(sdb);
    const f32x4 a0 = s4[(0 * TQB + d) * 16 + l16], a1 = s4[(1 * TQB + d) * 16 + l16];
    const f32x4 a2 = s4[(2 * TQB + d) * 16 + l16], a3 = s4[(3 * TQB + d) * 16 + l16];
    f32x4 v;
    v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
    v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
    *reinterpret_cast<f32x4*>(pr.db + q0 + d * 64 + (l16 << 2)) = v;
    ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq); ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
  }
  if (pr.partial != nullptr) {
    ssq = wave_sum64(ssq);
    __syncthreads();                       // every wave is done reading the parked tiles
    if (lane == 0) smem[wave] = ssq;
    __syncthreads();
    if (threadIdx.x == 0) pr.partial[tile_q * pr.tiles_p + tile_p] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  }
}

// ---- WGRAD, quadrants: no split-K (round 6 experiment) ---------------------------------------------------
// At minibatches <= 512 the reduction (the rows) is only 256-512 deep; wgrad_direct_body still splits it four ways over the waves
// and pays one LDS pass (64 KB parked, 256 KB read back per workgroup) to add the partial tiles.  Here each wave OWNS a 32 x 32
// quadrant of the 64 x 64 tile over the WHOLE reduction: four accumulators, 8-byte operand loads (two columns per lane), no
// parking, no barrier before the stores.  Same tile map, same db / sum-of-squares outputs (the partial is summed in another
// fixed order); every dW element is ONE chain over the rows instead of four quarter chains.
// Measured against wgrad_direct_body in the test harness (dqnhip_test_gemm mode 2, variant 3): profiles/r06_wgrad_quadrants.txt.
template <int NS = 8>
__device__ __forceinline__ void wgrad_quad_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wp = wave & 1, wq = wave >> 1;
  const int p0 = tile_p * 64 + wp * 32, q0 = tile_q * 64 + wq * 32;
  const int nst = pr.Kred >> 2;              // steps of 4 rows (lane group lg owns row 4 st + lg)
  const float* pp = pr.P + (size_t)lg * pr.ldp + p0 + li * 2;
  const float* qp = pr.Q + (size_t)lg * pr.ldq + q0 + li * 2;
  const size_t ldp4 = (size_t)4 * pr.ldp, ldq4 = (size_t)4 * pr.ldq;
  const bool want_db = (pr.db != nullptr) && (tile_p == 0) && (wp == 0);
  f32x4 acc[4];     // index qc * 2 + pc
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 dbacc = f32x2{0.f, 0.f};
  f32x2 rp[NS], rq[NS];
#define WQ_LOAD(slot, st) { rp[slot] = *reinterpret_cast<const f32x2*>(pp + (size_t)(st) * ldp4); rq[slot] = *reinterpret_cast<const f32x2*>(qp + (size_t)(st) * ldq4); }
#define WQ_COMPUTE(slot)                                                                \
  {                                                                                     \
    dbacc.x += rq[slot].x; dbacc.y += rq[slot].y;                                       \
    _Pragma("unroll") for (int qc = 0; qc < 2; ++qc)                                    \
    _Pragma("unroll") for (int pc = 0; pc < 2; ++pc)                                    \
        acc[qc * 2 + pc] = DQN_MFMA(rp[slot][pc], rq[slot][qc], acc[qc * 2 + pc]);      \
  }
  const int nstN = nst - nst % NS;
  if (nstN > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { WQ_LOAD(i, i) DQN_PIN(); }
    int st = 0;
    for (; st + NS < nstN; st += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { WQ_COMPUTE(i) DQN_PIN(); WQ_LOAD(i, st + NS + i) DQN_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { WQ_COMPUTE(i) }
  }
  for (int st = nstN; st < nst; ++st) { WQ_LOAD(0, st) WQ_COMPUTE(0) }
#undef WQ_LOAD
#undef WQ_COMPUTE
  // C/D map of acc[qc * 2 + pc]: lane (li, lg), register r = D[i = 4 lg + r][j = li] = dW[n = q0 + 2 li + qc][p = p0 + 2 (4 lg + r) + pc]
  float ssq = 0.0f;
#pragma unroll
  for (int qc = 0; qc < 2; ++qc) {
    const int n = q0 + li * 2 + qc;
    const f32x4 a0 = acc[qc * 2 + 0], a1 = acc[qc * 2 + 1];
    const f32x4 v0 = f32x4{a0.x, a1.x, a0.y, a1.y}, v1 = f32x4{a0.z, a1.z, a0.w, a1.w};     // p = p0 + 8 lg + 0..3, + 4..7
    ssq = fmaf(v0.x, v0.x, ssq); ssq = fmaf(v0.y, v0.y, ssq); ssq = fmaf(v0.z, v0.z, ssq); ssq = fmaf(v0.w, v0.w, ssq);
    ssq = fmaf(v1.x, v1.x, ssq); ssq = fmaf(v1.y, v1.y, ssq); ssq = fmaf(v1.z, v1.z, ssq); ssq = fmaf(v1.w, v1.w, ssq);
    float* c = pr.C + (size_t)n * pr.ldc + p0 + (lg << 3);
    *reinterpret_cast<f32x4*>(c) = v0; *reinterpret_cast<f32x4*>(c + 4) = v1;
  }
  if (want_db) {
    f32x2 v = dbacc;      // add the 4 lane groups (rows m + 0..3)
    v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
    v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
    if (lg == 0) {
      *reinterpret_cast<f32x2*>(pr.db + q0 + li * 2) = v;
      ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq);
    }
  }
  if (pr.partial != nullptr) {
    ssq = wave_sum64(ssq);
    if (lane == 0) smem[wave] = ssq;
    __syncthreads();
    if (threadIdx.x == 0) pr.partial[tile_q * pr.tiles_p + tile_p] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  }
}
template <int NS>
__global__ __launch_bounds__(256) void gemm_wgrad_quad(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  wgrad_quad_body<NS>(batch.prob[pi], tile_p, tile_q, smem);
}

// ---- WGRAD, narrow: 16 columns of dY per workgroup ------------------------------------------
// The first tower layer's dW_0[n][j] has only K_in = 64 / 128 columns j: with 64 x 64 tiles it is 16 / 32
// workgroups whose waves each hold 4.2 us of MFMA (a latency-bound 8 us launch on a sliver of the chip).
// Here a workgroup owns 16 outputs n (one dY column per lane, scalar loads) x 64 columns j: a quarter of
// the MFMA chain per wave, four times the workgroups.  Same K split over the 4 waves, same fixed-order
// reduction; db and the sum-of-squares partial (slot = tile_q * tiles_p + tile_p over 16-wide tiles).
template <int TPB>
__device__ __forceinline__ void wgrad_narrow_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NACC = TPB * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int p0 = tile_p * 64 * TPB, q0 = tile_q * 16;
  const int Kw = pr.Kred >> 2;
  const int nst = Kw >> 2;
  const float* pp = pr.P + (size_t)(wave * Kw + lg) * pr.ldp + p0 + li * 4;
  const float* qp = pr.Q + (size_t)(wave * Kw + lg) * pr.ldq + q0 + li;
  const size_t ldp = pr.ldp, ldq = pr.ldq;
  const bool want_db = (pr.db != nullptr) && (tile_p == 0);
  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.0f;
  constexpr int NS = kWgradRing;
  f32x4 rp[NS][TPB]; float rq[NS];
#define WN_LOAD(slot, st)                                                               \
  {                                                                                     \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
        rp[slot][b] = *reinterpret_cast<const f32x4*>(pp + (size_t)((st) << 2) * ldp + b * 64); \
    rq[slot] = qp[(size_t)((st) << 2) * ldq];                                           \
  }
#define WN_COMPUTE(slot)                                                                \
  {                                                                                     \
    dbacc += rq[slot];                                                                  \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                                    \
        acc[b * 4 + pc] = DQN_MFMA(rp[slot][b][pc], rq[slot], acc[b * 4 + pc]);         \
  }
  const int nstN = nst - nst % NS;
  if (nstN > 0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { WN_LOAD(i, i) DQN_PIN(); }
    int st = 0;
    for (; st + NS < nstN; st += NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) { WN_COMPUTE(i) DQN_PIN(); WN_LOAD(i, st + NS + i) DQN_PIN(); }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { WN_COMPUTE(i) }
  }
  for (int st = nstN; st < nst; ++st) { WN_LOAD(0, st) WN_COMPUTE(0) }
#undef WN_LOAD
#undef WN_COMPUTE

  park_accumulators<NACC>(smem, acc, wave, lane);
  float* sdb = smem + 4 * NACC * 64 * 4;      // [4 waves][16 li]
  if (want_db) {
    float v = dbacc;                          // add the 4 lane groups (rows m+0..3)
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (lg == 0) sdb[wave * 16 + li] = v;
  }
  __syncthreads();
  float ssq = 0.0f;
#pragma unroll
  for (int b = 0; b < TPB; ++b) {
    const f32x4 r0 = reduce_accumulator<NACC>(smem, b * 4 + 0, lane), r1 = reduce_accumulator<NACC>(smem, b * 4 + 1, lane);
    const f32x4 r2 = reduce_accumulator<NACC>(smem, b * 4 + 2, lane), r3 = reduce_accumulator<NACC>(smem, b * 4 + 3, lane);
    const int n = q0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (((b + r) & 3) == wave) {           // the four row groups of an accumulator set are shared out over the waves
        const int p = p0 + b * 64 + (lg << 4) + (r << 2);
        const f32x4 v = f32x4{r0[r], r1[r], r2[r], r3[r]};
        ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq); ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
        *reinterpret_cast<f32x4*>(pr.C + (size_t)n * pr.ldc + p) = v;
      }
    }
  }
  if (want_db && wave == 0 && lane < 16) {
    const float v = (sdb[0 * 16 + lane] + sdb[1 * 16 + lane]) + (sdb[2 * 16 + lane] + sdb[3 * 16 + lane]);
    pr.db[q0 + lane] = v;
    ssq = fmaf(v, v, ssq);
  }
  if (pr.partial != nullptr) {
    ssq = wave_sum64(ssq);
    __syncthreads();
    if (lane == 0) smem[wave] = ssq;
    __syncthreads();
    if (threadIdx.x == 0) pr.partial[tile_q * pr.tiles_p + tile_p] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  }
}

// ================================ FWD, coalesced =====================================
// Same tile / split-K structure as fwd_direct_body, but the k-contiguous operands are
// fetched as WHOLE 128-byte lines (8 rows x 128 B per wave instruction) and transposed into
// MFMA fragment layout through a wave-private LDS image — the fragment-shaped loads of
// fwd_direct_body (sixteen 64-B pieces per instruction) run the texture addresser at 1/4
// rate: measured 8.5 TB/s vs 20 TB/s for the same bytes (ablation v6/v9).
//   global (coalesced)  lane l -> row l>>3, 16-B chunk l&7          [2 loads / 16 rows / 32 k]
//   LDS image per 16-row block: [16 rows][8 chunks], chunk ^= row&7 (ds_write_b128: 8 lanes
//   of a row hit 8 distinct chunks; ds_read_b128 in MFMA layout is conflict-free, see
//   DESIGN.md for the lane-group check)
//   fragment            lane (i=l&15, g=l>>4), kb -> row i, chunk kb*4+g
// The LDS image is private to the wave (no barrier anywhere in the main loop); two images
// ping-pong, global loads run two 32-k steps ahead in registers.
// Requires Kred % 256 == 0 and Kred >= 512 (>= 4 steps of 32 k per wave, even count).
// WT (probe only, csrc/gemm_bench.hip): the output tile is stored write-through (`sc1`), the producer side of an
// in-launch hand-off without a release fence (guide G16 R1).  The learner instantiates WT = false.
template <int TP, int TQ, bool PIN, int NSLOT = 2, bool WT = false>
__device__ __forceinline__ void fwd_lds_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  constexpr int NB = TP + TQ;
  constexpr int NACC = TP * TQ;
  constexpr int SLOT = NB * 512;                 // floats per LDS image (NB blocks x 16 rows x 32 k)
  // NSLOT = 1: one image per wave.  LDS operations of one wave execute in issue order, so the next
  // step's ds_write cannot overtake this step's ds_read of the same image; half the LDS lets two
  // workgroups share a CU.
  constexpr int WSTR = (NSLOT * SLOT > NACC * 256) ? NSLOT * SLOT : NACC * 256;   // floats per wave region (images, later the parked tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int lr = lane >> 3, lc = lane & 7;
  const int p0 = tile_p * 16 * TP, q0 = tile_q * 16 * TQ;
  const int Kw = pr.Kred >> 2;
  const int T = Kw >> 5;                         // steps of 32 k
  float* wsm = smem + wave * WSTR;
  const float* gp[NB];
  size_t ld8[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b < TP) { gp[b] = pr.P + (size_t)(p0 + b * 16 + lr) * pr.ldp + wave * Kw + lc * 4; ld8[b] = (size_t)8 * pr.ldp; }
    else { gp[b] = pr.Q + (size_t)(q0 + (b - TP) * 16 + lr) * pr.ldq + wave * Kw + lc * 4; ld8[b] = (size_t)8 * pr.ldq; }
  }
  const int woff = lr * 32 + ((lc ^ lr) << 2);                 // + h*256 + b*512
  int roff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) roff[kb] = li * 32 + ((((kb << 2) + lg) ^ (li & 7)) << 2);

  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 G0[NB][2], G1[NB][2], F[NB][2], Fn[NB][2];

#define L_GLOAD(G, t)                                                                   \
  { _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                    \
      G[b][0] = *reinterpret_cast<const f32x4*>(gp[b] + ((t) << 5));                   \
      G[b][1] = *reinterpret_cast<const f32x4*>(gp[b] + ld8[b] + ((t) << 5)); } }
#define L_SWRITE(slot, G)                                                               \
  { _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                    \
      *reinterpret_cast<f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + woff) = G[b][0];        \
      *reinterpret_cast<f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + 256 + woff) = G[b][1]; } }
#define L_SREAD(FF, slot)                                                               \
  { _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                    \
      FF[b][0] = *reinterpret_cast<const f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + roff[0]); \
      FF[b][1] = *reinterpret_cast<const f32x4*>(wsm + ((slot) % NSLOT) * SLOT + b * 512 + roff[1]); } }
#define L_MFMA(FF)                                                                      \
  { _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                    \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                       \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
    _Pragma("unroll") for (int c = 0; c < TP; ++c)                                      \
        acc[a * TP + c] = DQN_MFMA(FF[c][kb][s], FF[TP + a][kb][s], acc[a * TP + c]); }
#define L_PIN() { if (PIN) DQN_PIN(); }
  // PIN: every half step {stage the next image: ds_write x2NB, global_load x2NB, ds_read x2NB | MFMA of the
  // current fragments} is one scheduling region whose staging instructions are spread through the MFMAs
  // in that order (hipcc on its own either sinks the loads to the end of the iteration — zero lookahead —
  // or, with plain order pinning, issues the 3 x 2NB staging instructions as a block with the MFMA pipe idle)
  constexpr int NOPS = 2 * NB, NMF = 8 * NACC;
  constexpr int MW = (NMF >= 5 * NOPS) ? 2 : 1, ML = (NMF >= 5 * NOPS) ? 2 : (NMF >= 2 * NOPS ? 1 : 0),
                MR = (NMF >= 3 * NOPS) ? 1 : 0;          // small tiles: fewer MFMAs than staging instructions
  constexpr bool SGB = PIN && (NMF >= NOPS);
#define L_SCHED_(HASLD)                                                                 \
  { if constexpr (SGB) {                                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < NOPS; ++i_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, MW, 0); } \
      if (HASLD) { _Pragma("unroll") for (int i_ = 0; i_ < NOPS; ++i_) {                \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); if (ML > 0) __builtin_amdgcn_sched_group_barrier(0x008, ML, 0); } } \
      _Pragma("unroll") for (int i_ = 0; i_ < NOPS; ++i_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); if (MR > 0) __builtin_amdgcn_sched_group_barrier(0x008, MR, 0); } \
      if (NMF - (MW + (HASLD ? ML : 0) + MR) * NOPS > 0)                                \
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - (MW + (HASLD ? ML : 0) + MR) * NOPS, 0); } }
#define L_SCHED() L_SCHED_(true)

  // (pinned prologue: the waitcnt pass merges the prologue's load order into the loop header, so an
  // interleaved prologue makes every in-loop vmcnt wait conservative)
  L_GLOAD(G0, 0) L_PIN() L_GLOAD(G1, 1) L_PIN()
  L_SWRITE(0, G0) L_PIN() L_GLOAD(G0, 2) L_PIN() L_SREAD(F, 0) L_PIN()
  int t = 0;
  for (; t + 4 < T; t += 2) {
    if constexpr (SGB) {
      L_SWRITE(1, G1) L_GLOAD(G1, t + 3) L_SREAD(Fn, 1) L_MFMA(F) L_SCHED() L_PIN()
      L_SWRITE(0, G0) L_GLOAD(G0, t + 4) L_SREAD(F, 0) L_MFMA(Fn) L_SCHED() L_PIN()
    } else {
      L_SWRITE(1, G1) L_PIN() L_GLOAD(G1, t + 3) L_PIN() L_SREAD(Fn, 1) L_PIN()
      L_MFMA(F) L_PIN()
      L_SWRITE(0, G0) L_PIN() L_GLOAD(G0, t + 4) L_PIN() L_SREAD(F, 0) L_PIN()
      L_MFMA(Fn) L_PIN()
    }
  }
  // t == T-4
  if constexpr (SGB) {
    L_SWRITE(1, G1) L_GLOAD(G1, T - 1) L_SREAD(Fn, 1) L_MFMA(F) L_SCHED() L_PIN()
    L_SWRITE(0, G0) L_SREAD(F, 0) L_MFMA(Fn) L_SCHED_(false) L_PIN()
    L_SWRITE(1, G1) L_SREAD(Fn, 1) L_MFMA(F) L_SCHED_(false) L_PIN()
    L_MFMA(Fn)
  } else {
    L_SWRITE(1, G1) L_GLOAD(G1, T - 1) L_SREAD(Fn, 1) L_PIN()
    L_MFMA(F) L_PIN()
    L_SWRITE(0, G0) L_SREAD(F, 0) L_PIN()
    L_MFMA(Fn) L_PIN()
    L_SWRITE(1, G1) L_SREAD(Fn, 1) L_PIN()
    L_MFMA(F) L_PIN()
    L_MFMA(Fn)
  }
#undef L_GLOAD
#undef L_SWRITE
#undef L_SREAD
#undef L_MFMA
#undef L_PIN
#undef L_SCHED
#undef L_SCHED_

  // this wave's bias pieces, requested ahead of the cross-wave reduction
  constexpr int NBV = (NACC + 3) / 4;
  f32x4 bvp[NBV], swp[NBV], dwp[NBV];
  if (pr.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) bvp[j] = *reinterpret_cast<const f32x4*>(pr.bias + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  if (pr.seed_w != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) swp[j] = *reinterpret_cast<const f32x4*>(pr.seed_w + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  if (pr.dot_w != nullptr) {
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int e = j * 4 + wave;
      if (e < NACC) dwp[j] = *reinterpret_cast<const f32x4*>(pr.dot_w + p0 + (e % TP) * 16 + (lg << 2));
    }
  }
  // park into this wave's own (now idle) staging region, reduce across waves in fixed order
  f32x4* park = reinterpret_cast<f32x4*>(wsm);
#pragma unroll
  for (int e = 0; e < NACC; ++e) park[e * 64 + lane] = acc[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NACC; ++e) {
    if ((e & 3) == wave) {
      const int a = e / TP, c = e % TP;
      const f32x4 a0 = reinterpret_cast<const f32x4*>(smem + 0 * WSTR)[e * 64 + lane];
      const f32x4 a1 = reinterpret_cast<const f32x4*>(smem + 1 * WSTR)[e * 64 + lane];
      const f32x4 a2 = reinterpret_cast<const f32x4*>(smem + 2 * WSTR)[e * 64 + lane];
      const f32x4 a3 = reinterpret_cast<const f32x4*>(smem + 3 * WSTR)[e * 64 + lane];
      f32x4 v;
      v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
      v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
      const int q = q0 + a * 16 + li, p = p0 + c * 16 + (lg << 2);
      if (pr.bias != nullptr) {
        const f32x4 bv = bvp[e >> 2];
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (pr.relu) { v.x = lrelu_fwd(v.x); v.y = lrelu_fwd(v.y); v.z = lrelu_fwd(v.z); v.w = lrelu_fwd(v.w); }
      if constexpr (WT) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pr.C, 0, pr.Qdim * pr.ldc * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)(((size_t)q * pr.ldc + p) * 4), 0, 16 /* sc1 */);
      } else {
        *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
      }
      if (pr.seed_w != nullptr) store_head_seed(pr, q, p, v, swp[e >> 2]);
      if (pr.dot_w != nullptr) store_head_dot(pr, q, p, lg, v, dwp[e >> 2]);
    }
  }
}

// ================================ DGRAD, coalesced ===================================
// dgrad_direct_body with the k-contiguous operand (dY, 16-row blocks) fetched as whole 128-B
// lines through the wave-private LDS transpose of fwd_lds_body; the weight operand (k-strided)
// stays a direct full-line load.  Requires Kred % 256 == 0 and Kred >= 512.
// s_rowscale (LDS, [16 TQ] floats, published before this body's barrier by a wave that does not run it — k_dgrad_qtrain): the
// reduced sums of row r are multiplied by s_rowscale[r] before the ReLU' mask (a dY panel whose rows share a late-known scalar factor)
// hook (k_dgrad_qtrain): after_prologue() runs once the operand pipeline is primed (a place to REQUEST data whose latency the
// main loop then hides), before_park() after the last MFMA and before this body's only barrier (a place to publish s_rowscale).
struct DgradNoHook { __device__ __forceinline__ void after_prologue() {} __device__ __forceinline__ void before_park() {} };
template <int TPB, int TQ, bool SCH = true, typename Hook = DgradNoHook>
__device__ __forceinline__ void dgrad_lds_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem, const float* s_rowscale, Hook& hook) {
  constexpr int NACC = TPB * 4 * TQ;
  constexpr int SLOT = TQ * 512;
  constexpr int WAVE_FLOATS = (2 * SLOT > NACC * 256) ? 2 * SLOT : NACC * 256;   // staging, later the parked tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int lr = lane >> 3, lc = lane & 7;
  const int p0 = tile_p * 64 * TPB, q0 = tile_q * 16 * TQ;
  const int Kw = pr.Kred >> 2;
  const int T = Kw >> 5;
  float* wsm = smem + wave * WAVE_FLOATS;
  const float* pp = pr.P + (size_t)(wave * Kw + lg * 4) * pr.ldp + p0 + li * 4;
  const size_t ldp = pr.ldp;
  const float* gq[TQ];
#pragma unroll
  for (int a = 0; a < TQ; ++a) gq[a] = pr.Q + (size_t)(q0 + a * 16 + lr) * pr.ldq + wave * Kw + lc * 4;
  const size_t ldq8 = (size_t)8 * pr.ldq;
  const int woff = lr * 32 + ((lc ^ lr) << 2);
  int roff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) roff[kb] = li * 32 + ((((kb << 2) + lg) ^ (li & 7)) << 2);

  // this lane's pieces of the ReLU' mask (one per (a,b): the r this wave owns in the epilogue),
  // requested before the reduction loop so that their latency is not exposed after it
  f32x4 mk[TQ * TPB];
  if (pr.mask != nullptr) {
#pragma unroll
    for (int ab = 0; ab < TQ * TPB; ++ab) {
      const int r = (wave - ab) & 3;
      mk[ab] = *reinterpret_cast<const f32x4*>(pr.mask + (size_t)(q0 + (ab / TPB) * 16 + li) * pr.ldm + p0 + (ab % TPB) * 64 + (lg << 4) + (r << 2));
    }
  }
  f32x4 acc[NACC];
#pragma unroll
  for (int e = 0; e < NACC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 G0[TQ][2], G1[TQ][2], F[TQ][2], Fn[TQ][2];
  f32x4 P0[2][4][TPB], P1[2][4][TPB];

#define D_GLOADQ(G, t)                                                                  \
  { _Pragma("unroll") for (int a = 0; a < TQ; ++a) {                                    \
      G[a][0] = *reinterpret_cast<const f32x4*>(gq[a] + ((t) << 5));                   \
      G[a][1] = *reinterpret_cast<const f32x4*>(gq[a] + ldq8 + ((t) << 5)); } }
#define D_GLOADP(PP, t)                                                                 \
  { _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                    \
    _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                    \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
        PP[kb][s2][b] = *reinterpret_cast<const f32x4*>(pp + (size_t)(((t) << 5) + (kb << 4) + s2) * ldp + b * 64); }
#define D_SWRITE(slot, G)                                                               \
  { _Pragma("unroll") for (int a = 0; a < TQ; ++a) {                                    \
      *reinterpret_cast<f32x4*>(wsm + (slot) * SLOT + a * 512 + woff) = G[a][0];        \
      *reinterpret_cast<f32x4*>(wsm + (slot) * SLOT + a * 512 + 256 + woff) = G[a][1]; } }
#define D_SREAD(FF, slot)                                                               \
  { _Pragma("unroll") for (int a = 0; a < TQ; ++a) {                                    \
      FF[a][0] = *reinterpret_cast<const f32x4*>(wsm + (slot) * SLOT + a * 512 + roff[0]); \
      FF[a][1] = *reinterpret_cast<const f32x4*>(wsm + (slot) * SLOT + a * 512 + roff[1]); } }
#define D_MFMA(FF, PP)                                                                  \
  { _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                    \
    _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                    \
    _Pragma("unroll") for (int a = 0; a < TQ; ++a)                                      \
    _Pragma("unroll") for (int b = 0; b < TPB; ++b)                                     \
    _Pragma("unroll") for (int pc = 0; pc < 4; ++pc)                                    \
        acc[(a * TPB + b) * 4 + pc] = DQN_MFMA(PP[kb][s2][b][pc], FF[a][kb][s2], acc[(a * TPB + b) * 4 + pc]); }

  // One scheduling region per half step, staging instructions spread through the MFMAs in this order
  // (see fwd_lds_body): dY image ds_writes, next dY loads, fragment ds_reads among the first MFMAs;
  // the W loads that refill the P registers among the second half (their registers are free once the
  // kb = 0 MFMAs have issued).  Pinned prologue so that the in-loop vmcnt waits are counted, not 0.
  constexpr int NQ = 2 * TQ, NPL = 8 * TPB, NMF = 32 * TQ * TPB, HALF = NMF / 2;
#define D_PIN() { if (SCH) DQN_PIN(); }
#define D_SCHED_(HASQ, HASP)                                                            \
  if constexpr (SCH) { _Pragma("unroll") for (int i_ = 0; i_ < NQ; ++i_) {                                 \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } \
    if (HASQ) { _Pragma("unroll") for (int i_ = 0; i_ < NQ; ++i_) {                     \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } } \
    _Pragma("unroll") for (int i_ = 0; i_ < NQ; ++i_) {                                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, HALF - (HASQ ? 3 : 2) * NQ, 0);         \
    if (HASP) { _Pragma("unroll") for (int i_ = 0; i_ < NPL; ++i_) {                    \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, HALF / NPL, 0); } } \
    else __builtin_amdgcn_sched_group_barrier(0x008, HALF, 0);                          \
    DQN_PIN(); }

  D_GLOADQ(G0, 0) D_PIN() D_GLOADQ(G1, 1) D_PIN() D_GLOADP(P0, 0) D_PIN()
  hook.after_prologue(); D_PIN()      // (with the first, cold round of operand requests: later, its cold misses hold up the in-order vmcnt of the loop's loads)
  D_SWRITE(0, G0) D_PIN() D_GLOADQ(G0, 2) D_PIN() D_SREAD(F, 0) D_PIN() D_GLOADP(P1, 1) D_PIN()
  int t = 0;
  for (; t + 4 < T; t += 2) {
    D_SWRITE(1, G1) D_GLOADQ(G1, t + 3) D_SREAD(Fn, 1) D_MFMA(F, P0) D_GLOADP(P0, t + 2) D_SCHED_(true, true)
    D_SWRITE(0, G0) D_GLOADQ(G0, t + 4) D_SREAD(F, 0) D_MFMA(Fn, P1) D_GLOADP(P1, t + 3) D_SCHED_(true, true)
  }
  // t == T-4
  D_SWRITE(1, G1) D_GLOADQ(G1, T - 1) D_SREAD(Fn, 1) D_MFMA(F, P0) D_GLOADP(P0, T - 2) D_SCHED_(true, true)
  D_SWRITE(0, G0) D_SREAD(F, 0) D_MFMA(Fn, P1) D_GLOADP(P1, T - 1) D_SCHED_(false, true)
  D_SWRITE(1, G1) D_SREAD(Fn, 1) D_MFMA(F, P0) D_SCHED_(false, false)
  D_MFMA(Fn, P1)
  D_PIN() hook.before_park();
#undef D_SCHED_
#undef D_PIN
#undef D_GLOADQ
#undef D_GLOADP
#undef D_SWRITE
#undef D_SREAD
#undef D_MFMA

  f32x4* park = reinterpret_cast<f32x4*>(wsm);
#pragma unroll
  for (int e = 0; e < NACC; ++e) park[e * 64 + lane] = acc[e];
  __syncthreads();
  auto red = [&](int e) {
    const f32x4 a0 = reinterpret_cast<const f32x4*>(smem + 0 * WAVE_FLOATS)[e * 64 + lane];
    const f32x4 a1 = reinterpret_cast<const f32x4*>(smem + 1 * WAVE_FLOATS)[e * 64 + lane];
    const f32x4 a2 = reinterpret_cast<const f32x4*>(smem + 2 * WAVE_FLOATS)[e * 64 + lane];
    const f32x4 a3 = reinterpret_cast<const f32x4*>(smem + 3 * WAVE_FLOATS)[e * 64 + lane];
    f32x4 v;
    v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
    v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
    return v;
  };
#pragma unroll
  for (int ab = 0; ab < TQ * TPB; ++ab) {
    // with TQ*TPB < 4 the four pc-accumulator groups of one (a,b) are shared out over the waves by r
    const int a = ab / TPB, b = ab % TPB;
    const f32x4 r0 = red(ab * 4 + 0), r1 = red(ab * 4 + 1), r2 = red(ab * 4 + 2), r3 = red(ab * 4 + 3);
    const int q = q0 + a * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (((ab + r) & 3) == wave) {
        const int p = p0 + b * 64 + (lg << 4) + (r << 2);
        f32x4 v = f32x4{r0[r], r1[r], r2[r], r3[r]};
        if (s_rowscale != nullptr) { const float sc = s_rowscale[a * 16 + li]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        if (pr.mask != nullptr) {
          const f32x4 mv = mk[ab];
          v.x *= lrelu_mask(mv.x); v.y *= lrelu_mask(mv.y); v.z *= lrelu_mask(mv.z); v.w *= lrelu_mask(mv.w);
        }
        *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
      }
    }
  }
}

// ---- kernels: thin wrappers over the bodies --------------------------------------------
template <int TP, int TQ>
__global__ __launch_bounds__(256) void gemm_fwd_direct(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  fwd_direct_body<TP, TQ>(batch.prob[pi], tile_p, tile_q, smem);
}
template <int TP, int TQ, bool PIN, int NSLOT = 2>
__global__ __launch_bounds__(256) void gemm_fwd_lds(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  fwd_lds_body<TP, TQ, PIN, NSLOT>(batch.prob[pi], tile_p, tile_q, smem);
}
template <int TPB, int TQ>
__global__ __launch_bounds__(256) void gemm_dgrad_direct(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  dgrad_direct_body<TPB, TQ>(batch.prob[pi], tile_p, tile_q, smem);
}
template <int TPB, int TQ, bool SCH = true>
__device__ __forceinline__ void dgrad_lds_body(const GemmProblem& pr, int tile_p, int tile_q, float* smem) {
  DgradNoHook none;
  dgrad_lds_body<TPB, TQ, SCH, DgradNoHook>(pr, tile_p, tile_q, smem, nullptr, none);
}
template <int TPB, int TQ>
__global__ __launch_bounds__(256) void gemm_dgrad_lds(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  dgrad_lds_body<TPB, TQ>(batch.prob[pi], tile_p, tile_q, smem);
}
template <int TPB, int TQB>
__global__ __launch_bounds__(256) void gemm_wgrad_direct(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  wgrad_direct_body<TPB, TQB>(batch.prob[pi], tile_p, tile_q, smem);
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void gemm_dgrad_narrow(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  dgrad_narrow_body(batch.prob[pi], tile_p, tile_q, smem);
}
// q(s, mu(s)) = q_values(critic tower top) and its avg-Q partials (src/dqn.cpp:913-916) as RIDER blocks of the narrow dgrad
// launch: a handful of 16 x 16 tiles (16 workgroups at 256 rows) that leaves most of the chip idle.  Nothing on the
// backward chain reads q — only the update's statistics do — so it needs no launch of its own (it used to ride in the
// dq = -1 head-backward launch, which is gone: the seed comes out of the top layer's forward epilogue, GemmProblem::seed_w).
// One wave per row, k-strips of float4; the riders come LAST in the grid.
struct QHeadRider {
  const float* X4; const float* W; const float* bias;   // tower top [rows][H], head weights [H], head bias [1]
  float* q_out; double* qsum_partial;                   // [rows] each
  int H, rows, blocks;                                  // blocks = ceil(rows / 4) (0: none)
  const _Float16* X416;                                 // fp16 learner: the tower top in fp16 (then X4 is null); last member (aggregate initialisers of the fp32 call sites leave it null)
};
__device__ __forceinline__ void q_head_rider(const QHeadRider& r, const int blk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blk * 4 + wave;
  if (row >= r.rows) return;
  const size_t x0 = (size_t)row * r.H;
  float acc = 0.0f;
  if (r.X416 != nullptr) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    for (int k = lane * 4; k < r.H; k += 256) {
      const h4 xh = *reinterpret_cast<const h4*>(r.X416 + x0 + k); const f32x4 wv = *reinterpret_cast<const f32x4*>(r.W + k);
      acc = fmaf((float)xh.x, wv.x, acc); acc = fmaf((float)xh.y, wv.y, acc); acc = fmaf((float)xh.z, wv.z, acc); acc = fmaf((float)xh.w, wv.w, acc);
    }
  } else
  for (int k = lane * 4; k < r.H; k += 256) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(r.X4 + x0 + k), wv = *reinterpret_cast<const f32x4*>(r.W + k);
    acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
  }
  acc = wave_sum64(acc);
  if (lane == 0) { const float v = acc + r.bias[0]; r.q_out[row] = v; r.qsum_partial[row] = (double)v; }
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void gemm_dgrad_narrow_qrider(const GemmBatch batch, const QHeadRider rider) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= batch.total_tiles) { q_head_rider(rider, (int)blockIdx.x - batch.total_tiles); return; }
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  dgrad_narrow_body(batch.prob[pi], tile_p, tile_q, smem);
}
template <int TPB>
__global__ __launch_bounds__(256) void gemm_wgrad_narrow(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  wgrad_narrow_body<TPB>(batch.prob[pi], tile_p, tile_q, smem);
}
// One layer's backward in ONE launch: problems with mode GEMM_DGRAD (64x16 tiles) and
// GEMM_WGRAD (64x64 tiles) side by side.  dX_{l-1} = dZ_l W_l and dW_l = dZ_l^T X_{l-1} only
// share their input dZ_l, so a 256x1024x1024 layer offers 256 + 256 workgroups = 2 per CU.
template <int TQD = 1, bool DLDS = false>
__global__ __launch_bounds__(256) void gemm_bwd_pair_direct(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  const GemmProblem& pr = batch.prob[pi];
  if (pr.mode == GEMM_WGRAD) wgrad_direct_body<1, 1>(pr, tile_p, tile_q, smem);
  else if constexpr (DLDS) dgrad_lds_body<1, TQD, false>(pr, tile_p, tile_q, smem);   // (scheduled form measured 0.5 us slower beside the co-resident wgrad wave)
  else dgrad_direct_body<1, TQD>(pr, tile_p, tile_q, smem);
}

// The same layer backward with ONE workgroup type: workgroup b computes wgrad tile b and then dgrad
// tile b in one instruction stream (prob[0] = dgrad, prob[1] = wgrad).  Two co-resident workgroups
// per CU measured ~ the SUM of their stand-alone times (the pair kernel above: 15.5 us for 2 x 4.2 us
// of MFMA); one workgroup doing both pays the per-launch fixed cost once and keeps one wave per SIMD.
template <bool DLDS>
__device__ __forceinline__ void bwd_seq_block(const GemmBatch& batch, const int b, float* smem) {
  int tile_p, tile_q;
  // wgrad first (measured 14.9 us; dgrad first 15.3, also with the wgrad ring pre-issued under the
  // dgrad epilogue; round 5: wgrad first with the dgrad tile's first operand requests issued BEFORE the wgrad tile parks /
  // reduces / stores — 15.0-15.1 against 14.7 us; the same requests issued before the WHOLE wgrad tile — 14.9: three cold loads
  // in flight do not pay for the registers and wait counts they hold through the other tile)
  const GemmProblem& pw = batch.prob[1];
  if (b < pw.tiles_p * pw.tiles_q) {
    tile_of_problem(pw, b, tile_p, tile_q);
    wgrad_direct_body<1, 1>(pw, tile_p, tile_q, smem);
  }
  __syncthreads();
  const GemmProblem& pd = batch.prob[0];
  if (b < pd.tiles_p * pd.tiles_q) {
    tile_of_problem(pd, b, tile_p, tile_q);
    if constexpr (DLDS) dgrad_lds_body<1, 1, true>(pd, tile_p, tile_q, smem);
    else dgrad_direct_body<1, 1>(pd, tile_p, tile_q, smem);
  }
}
template <bool DLDS>
__global__ __launch_bounds__(256) void gemm_bwd_seq(const GemmBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_seq_block<DLDS>(batch, (int)blockIdx.x, smem);
}

// The head layer's weight / bias gradients (dWh[j][k] = sum_m dYh[m][j] X4[m][k], dbh[j] = sum_m dYh[m][j]) as RIDER blocks of a
// later launch of the same net's backward.  The head-backward kernel produces dZ for that launch and used to produce dWh too — through
// row-chunk slabs and an arrival counter whose tail (drain, barrier, counter, barrier, slab reads) was 2.5 us per launch at the end
// of a 5-8 us kernel.  Nothing before the optimiser pass reads dWh, and a rider block (8 columns x 32 row groups, every row of its
// columns: no cross-block reduction) is done in ~3 us.  dy: the head diffs the head-backward kernel consumed (critic: dq [rows]; actor: the post-invert diffs [rows][16]).
struct HeadWgradRider {
  const float* dy; int lddy;
  const float* X4; int H, rows;
  float* dW; float* db; float* partial;     // [NH][H], [NH], one sum-of-squares slot per rider block (H / 16)
  int blocks;                               // H / kRiderCW rider blocks, FIRST in the grid (0: none)
};
constexpr int kRiderCW = 8;                    // columns per rider block (x 32 row groups)
template <int NH>
__device__ __forceinline__ void head_wgrad_rider(const HeadWgradRider& r, const int blk, float* smem) {
  constexpr int CW = kRiderCW, RG = 256 / CW;
  const int tid = threadIdx.x, kc = tid % CW, rg = tid / CW;
  const int k = blk * CW + kc;
  float* s_dy = smem;                          // [rows][NH]
  float* s_acc = smem + r.rows * NH;           // [RG][NH][CW]
  const int per = (r.rows + RG - 1) / RG, m0 = rg * per, m1 = m0 + per < r.rows ? m0 + per : r.rows;
  constexpr int RB = 8;                        // every row of a 256-row minibatch in flight at once: the tower top was written
                                               // many launches ago (Infinity Cache / HBM latency, not L2)
  float xpre[RB];
#pragma unroll
  for (int u = 0; u < RB; ++u) xpre[u] = (m0 + u < m1) ? r.X4[(size_t)(m0 + u) * r.H + k] : 0.0f;
  for (int i = tid; i < r.rows * NH; i += 256) s_dy[i] = r.dy[(size_t)(i / NH) * r.lddy + (i % NH)];
  __syncthreads();
  float acc[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) acc[j] = 0.0f;
  for (int mb = m0; mb < m1; mb += RB) {
    float xb[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) xb[u] = (mb == m0) ? xpre[u] : ((mb + u < m1) ? r.X4[(size_t)(mb + u) * r.H + k] : 0.0f);
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int m = mb + u;
      if (m >= m1) break;
#pragma unroll
      for (int j = 0; j < NH; ++j) acc[j] = fmaf(s_dy[m * NH + j], xb[u], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NH; ++j) s_acc[(rg * NH + j) * CW + kc] = acc[j];
  __syncthreads();
  float ssq = 0.0f;
  if (tid < NH * CW) {                         // row groups added in index order
    const int j = tid / CW, c = tid % CW;
    float v = 0.0f;
#pragma unroll 8
    for (int g = 0; g < RG; ++g) v += s_acc[(g * NH + j) * CW + c];
    r.dW[(size_t)j * r.H + blk * CW + c] = v;
    ssq = v * v;
  } else if (blk == 0 && tid < NH * CW + NH) { // bias gradient: rows in index order
    const int j = tid - NH * CW;
    float v = 0.0f;
    for (int m = 0; m < r.rows; ++m) v += s_dy[m * NH + j];
    r.db[j] = v;
    ssq = v * v;
  }
  ssq = wave_sum64(ssq);
  __syncthreads();
  if ((tid & 63) == 0) s_acc[tid >> 6] = ssq;
  __syncthreads();
  if (tid == 0 && r.partial != nullptr) r.partial[blk] = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
}
// The carrier is the FIRST tower layer's narrow wgrad launch (the last launch before the optimiser pass): 64-128 tiles, so the rider
// blocks land on CUs of their own instead of beside a GEMM wave (as riders of the top layer's gemm_bwd_seq: +0.9 us on that launch)
template <int NH>
__global__ __launch_bounds__(256) void gemm_wgrad_narrow_rider(const GemmBatch batch, const HeadWgradRider rider) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < rider.blocks) { head_wgrad_rider<NH>(rider, (int)blockIdx.x, smem); return; }
  int pi, tile_p, tile_q;
  const int blk = (int)blockIdx.x - rider.blocks;
  pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i) if (i < batch.n && blk >= batch.prob[i].tile_base) pi = i;
  tile_of_problem(batch.prob[pi], blk - batch.prob[pi].tile_base, tile_p, tile_q);
  wgrad_narrow_body<1>(batch.prob[pi], tile_p, tile_q, smem);
}

// The LAST launch of a net's backward under the shifted schedule (learner.hip tower_backward): layer 1's wgrad (prob[0]:
// 64 x 64 tiles), the first layer's narrow wgrad (prob[1]: 16-output tiles) and the head's dW / db rider blocks.  The
// first layer's wgrad alone is 64-128 short workgroups — a 6-us launch of launch floor; beside a full wgrad it costs ~1.
// Long workgroups first in the grid.
template <int NH>
__global__ __launch_bounds__(256) void gemm_wgrad_tail(const GemmBatch batch, const HeadWgradRider rider, const TailsArgs tails) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int blk = (int)blockIdx.x;
  int tile_p, tile_q;
  if (tails.on && blk == (int)gridDim.x - 1) { tails_block(tails, smem, reinterpret_cast<double*>(smem + 8)); return; }   // data-parallel learners: one more block, last in the grid
  if (blk < rider.blocks) { head_wgrad_rider<NH>(rider, blk, smem); return; }
  blk -= rider.blocks;
  const GemmProblem& p0 = batch.prob[1];
  const int n0 = p0.tiles_p * p0.tiles_q;
  if (blk < n0) { tile_of_problem(p0, blk, tile_p, tile_q); wgrad_narrow_body<1>(p0, tile_p, tile_q, smem); return; }
  blk -= n0;
  const GemmProblem& p1 = batch.prob[0];
  tile_of_problem(p1, blk, tile_p, tile_q); wgrad_direct_body<1, 1>(p1, tile_p, tile_q, smem);
}

// ---- launchers ------------------------------------------------------------------------

// When set (by the learner's timing mode), the next launch is bracketed by these events through
// hipExtLaunchKernelGGL: they carry the dispatch packet's own start/stop timestamps, i.e. the
// same kernel duration rocprofv3 reports, without the cost of separate event records.
struct LaunchTimer { hipEvent_t start = nullptr, stop = nullptr; };
inline LaunchTimer& launch_timer() { static thread_local LaunchTimer t; return t; }

template <typename K>
inline hipError_t direct_launch(K kernel, GemmBatch& batch, int BP, int BQ, int lds_bytes, hipStream_t stream) {
  int base = 0;
  for (int i = 0; i < batch.n; ++i) {
    GemmProblem& p = batch.prob[i];
    p.tiles_p = p.Pdim / BP;
    p.tiles_q = p.Qdim / BQ;
    p.tile_base = base;
    base += p.tiles_p * p.tiles_q;
  }
  batch.total_tiles = base;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL(kernel, dim3(base), dim3(256), lds_bytes, stream, lt.start, lt.stop, 0, batch); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL(kernel, dim3(base), dim3(256), lds_bytes, stream, batch);
  return hipGetLastError();
}

template <int TP, int TQ>
inline hipError_t fwd_direct_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_fwd_direct<TP, TQ>, b, 16 * TP, 16 * TQ, 4 * TP * TQ * 64 * 16, s);
}
template <int TP, int TQ, bool PIN, int NSLOT = 2>
constexpr int fwd_lds_bytes() {
  return 4 * 4 * ((NSLOT * (TP + TQ) * 512 > TP * TQ * 256) ? NSLOT * (TP + TQ) * 512 : TP * TQ * 256);
}
template <int TP, int TQ, bool PIN, int NSLOT = 2>
inline hipError_t fwd_lds_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_fwd_lds<TP, TQ, PIN, NSLOT>, b, 16 * TP, 16 * TQ, (fwd_lds_bytes<TP, TQ, PIN, NSLOT>()), s);
}
template <int TPB, int TQ>
inline hipError_t dgrad_direct_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_dgrad_direct<TPB, TQ>, b, 64 * TPB, 16 * TQ, 4 * TPB * 4 * TQ * 64 * 16, s);
}
inline hipError_t dgrad_direct32_launch(GemmBatch& b, hipStream_t s) { return direct_launch(gemm_dgrad_direct32<0>, b, 32, 16, 4 * 2 * 64 * 16, s); }
template <int TPB, int TQ>
constexpr int dgrad_lds_bytes() { return 4 * ((2 * TQ * 512 > TPB * 4 * TQ * 256) ? 2 * TQ * 512 : TPB * 4 * TQ * 256) * 4; }
template <int TPB, int TQ>
inline hipError_t dgrad_lds_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_dgrad_lds<TPB, TQ>, b, 64 * TPB, 16 * TQ, dgrad_lds_bytes<TPB, TQ>(), s);
}
template <int TPB, int TQB>
inline hipError_t wgrad_direct_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_wgrad_direct<TPB, TQB>, b, 64 * TPB, 64 * TQB,
                       4 * TPB * 4 * TQB * 4 * 64 * 16 + 4 * TQB * 16 * 16, s);
}
template <int NS>
inline hipError_t wgrad_quad_launch(GemmBatch& b, hipStream_t s) { return direct_launch(gemm_wgrad_quad<NS>, b, 64, 64, 64, s); }
// mixed dgrad(64x16)/wgrad(64x64) launch; every problem carries its own mode
template <int TQD, bool DLDS = false>
inline hipError_t bwd_pair_direct_launch(GemmBatch& batch, hipStream_t stream) {
  int base = 0;
  for (int i = 0; i < batch.n; ++i) {
    GemmProblem& p = batch.prob[i];
    p.tiles_p = p.Pdim / 64;
    p.tiles_q = p.Qdim / (p.mode == GEMM_WGRAD ? 64 : 16 * TQD);
    p.tile_base = base;
    base += p.tiles_p * p.tiles_q;
  }
  batch.total_tiles = base;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL((gemm_bwd_pair_direct<TQD, DLDS>), dim3(base), dim3(256), 4 * 16 * 64 * 16 + 4 * 16 * 16, stream, lt.start, lt.stop, 0, batch); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL((gemm_bwd_pair_direct<TQD, DLDS>), dim3(base), dim3(256), 4 * 16 * 64 * 16 + 4 * 16 * 16, stream, batch);
  return hipGetLastError();
}
inline hipError_t dgrad_narrow_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_dgrad_narrow<0>, b, 16, 16, 4 * 64 * 16, s);
}
inline hipError_t dgrad_narrow_qrider_launch(GemmBatch& batch, const QHeadRider& rider, hipStream_t stream) {
  int base = 0;
  for (int i = 0; i < batch.n; ++i) {
    GemmProblem& p = batch.prob[i];
    p.tiles_p = p.Pdim / 16; p.tiles_q = p.Qdim / 16; p.tile_base = base;
    base += p.tiles_p * p.tiles_q;
  }
  batch.total_tiles = base;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL(gemm_dgrad_narrow_qrider<0>, dim3(base + rider.blocks), dim3(256), 4 * 64 * 16, stream, lt.start, lt.stop, 0, batch, rider); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL(gemm_dgrad_narrow_qrider<0>, dim3(base + rider.blocks), dim3(256), 4 * 64 * 16, stream, batch, rider);
  return hipGetLastError();
}
template <int TPB>
inline hipError_t wgrad_narrow_launch(GemmBatch& b, hipStream_t s) {
  return direct_launch(gemm_wgrad_narrow<TPB>, b, 64 * TPB, 16, 4 * TPB * 4 * 64 * 16 + 4 * 16 * 4, s);
}
template <int NH>
inline hipError_t wgrad_narrow_rider_launch(GemmBatch& batch, const HeadWgradRider& rider, hipStream_t stream) {
  int base = 0;
  for (int i = 0; i < batch.n; ++i) {
    GemmProblem& p = batch.prob[i];
    p.tiles_p = p.Pdim / 64; p.tiles_q = p.Qdim / 16; p.tile_base = base;
    base += p.tiles_p * p.tiles_q;
  }
  batch.total_tiles = base;
  const size_t need = (size_t)(rider.rows * NH + 16 * NH * 16) * sizeof(float);
  const size_t lds = need > (size_t)(4 * 4 * 64 * 16 + 4 * 16 * 4) ? need : (size_t)(4 * 4 * 64 * 16 + 4 * 16 * 4);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL((gemm_wgrad_narrow_rider<NH>), dim3(base + rider.blocks), dim3(256), lds, stream, lt.start, lt.stop, 0, batch, rider); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL((gemm_wgrad_narrow_rider<NH>), dim3(base + rider.blocks), dim3(256), lds, stream, batch, rider);
  return hipGetLastError();
}
// prob[0]: wgrad on 64 x 64 tiles, prob[1]: narrow wgrad on 64 x 16 tiles; rider.blocks may be 0
template <int NH>
inline hipError_t wgrad_tail_launch(GemmBatch& batch, const HeadWgradRider& rider, hipStream_t stream, const TailsArgs* tails_in = nullptr) {
  TailsArgs tails{}; if (tails_in != nullptr) { tails = *tails_in; tails.on = 1; }
  GemmProblem& w1 = batch.prob[0]; GemmProblem& w0 = batch.prob[1];
  w1.tiles_p = w1.Pdim / 64; w1.tiles_q = w1.Qdim / 64; w1.tile_base = 0;
  w0.tiles_p = w0.Pdim / 64; w0.tiles_q = w0.Qdim / 16; w0.tile_base = w1.tiles_p * w1.tiles_q;
  const int grid = w0.tile_base + w0.tiles_p * w0.tiles_q + rider.blocks + (tails.on ? 1 : 0);
  batch.total_tiles = grid;
  const size_t need = rider.blocks ? (size_t)(rider.rows * NH + 16 * NH * 16) * sizeof(float) : 0;
  const size_t lds = std::max(need, (size_t)(4 * 16 * 64 * 16 + 4 * 16 * 16));
  if (lds > 80 * 1024) return hipErrorInvalidValue;          // two workgroups per CU
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL((gemm_wgrad_tail<NH>), dim3(grid), dim3(256), lds, stream, lt.start, lt.stop, 0, batch, rider, tails); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL((gemm_wgrad_tail<NH>), dim3(grid), dim3(256), lds, stream, batch, rider, tails);
  return hipGetLastError();
}
template <bool DLDS>
inline hipError_t bwd_seq_launch(GemmBatch& batch, hipStream_t stream) {
  // prob[0] dgrad (64 x 16 tiles), prob[1] wgrad (64 x 64 tiles)
  GemmProblem& d = batch.prob[0]; GemmProblem& w = batch.prob[1];
  d.tiles_p = d.Pdim / 64; d.tiles_q = d.Qdim / 16; d.tile_base = 0;
  w.tiles_p = w.Pdim / 64; w.tiles_q = w.Qdim / 64; w.tile_base = 0;
  const int nd = d.tiles_p * d.tiles_q, nw = w.tiles_p * w.tiles_q;
  const int grid = nd > nw ? nd : nw;
  batch.total_tiles = grid;
  constexpr int lds = 4 * 16 * 64 * 16 + 4 * 16 * 16;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL((gemm_bwd_seq<DLDS>), dim3(grid), dim3(256), lds, stream, lt.start, lt.stop, 0, batch); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL((gemm_bwd_seq<DLDS>), dim3(grid), dim3(256), lds, stream, batch);
  return hipGetLastError();
}
template <typename K>
inline hipError_t direct_prepare(K kernel, int lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

}  // namespace dqnhip
