// gemm_mfma.hip.h — the FIRST (LDS-staged, one barrier per K tile) fp32 MFMA GEMM family.
// TEST-ONLY: kept as the A/B baseline of dqnhip_test_gemm (libdqnhip_test.so); the learner
// launches the gemm_direct.hip.h family and this header is not part of libdqnhip.so.
//
// Replaces the Caffe InnerProduct forward/backward GEMMs the reference reaches
// from src/dqn.cpp:751, 904, 923, 963, 1013, with the
// ReLU(negative_slope=0.01) forward/backward (src/dqn.cpp:292-301) fused into
// the epilogues.
//
// One kernel template, three modes.  Every mode computes C[q][p] = sum_k
// Pop(p,k) * Qop(q,k) with `p` the contiguous output dimension:
//
//   FWD   Y[m][n]  = lrelu(sum_k X[m][k] W[n][k] + b[n])      P=W (KC)  Q=X  (KC)
//   DGRAD dX[m][j] = (sum_n dY[m][n] W[n][j]) * lrelu'(Xp[m][j]) P=W (KS)  Q=dY (KC)
//   WGRAD dW[n][j] = sum_m dY[m][n] X[m][j] ; db[n] = sum_m dY[m][n]
//                                                             P=X (KS)  Q=dY (KS)
//
// Operand kinds (how the reduction index k lies in memory):
//   KC  "k-contiguous": row = free index, k runs along the row.  LDS image:
//       [rows][64 floats], 16-byte slots XOR-swizzled with (row & 15) so the
//       ds_read_b128 fragment reads are bank-conflict free (guide §6 G4).
//   KS  "k-strided": row = k, free index runs along the row.  LDS image:
//       [64 k-rows][cols + 4]; the +4 pad makes (4*stride) % 32 == 16 so the
//       ds_read_b32 fragment reads of a 32-lane half hit 32 distinct banks.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fmaf chain per
// instruction).  The P fragment is the MFMA A operand and the Q fragment the B
// operand, so a lane ends up holding 4 consecutive p of one q: one float4
// store per accumulator.  Within a 16-wide k block the lane group g = lane>>4
// owns k = kb*16 + g*4 + s at step s — a fixed permutation of the summation
// order, identical for every launch (deterministic, no atomics, no split-K).
//
// Tiling: 256 threads = 4 waves arranged WP x WQ over a BP x BQ output tile,
// BK = 64 per stage, two LDS stages, global->register->LDS staging with the
// next tile's global loads issued before the current tile's MFMAs (guide T14).
// A launch may carry up to 4 independent problems (grouped GEMM) so that the
// small-M layers of two networks fill the 256 CUs together.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_common.hip.h"

namespace dqnhip {

template <int MODE, int BP, int BQ, int WP, int WQ>
struct GemmCfg {
  static constexpr bool P_KC = (MODE == GEMM_FWD);
  static constexpr bool Q_KC = (MODE != GEMM_WGRAD);
  static constexpr int BK = 64;
  static constexpr int TP = BP / WP / 16;   // 16x16 accumulators per wave along p
  static constexpr int TQ = BQ / WQ / 16;
  static constexpr int P_FLOATS = P_KC ? BP * 64 : 64 * (BP + 4);
  static constexpr int Q_FLOATS = Q_KC ? BQ * 64 : 64 * (BQ + 4);
  static constexpr int STAGE_FLOATS = P_FLOATS + Q_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
  static constexpr int NP4 = BP / 16;       // float4 per thread per P tile
  static constexpr int NQ4 = BQ / 16;
  static_assert(WP * WQ == 4, "4 waves per workgroup");
  static_assert(BP % (16 * WP) == 0 && BQ % (16 * WQ) == 0, "wave tile");
};

// ---- staging helpers -------------------------------------------------------

// KC tile: R rows x 64 k.  thread f4 index f: row = f>>4, slot = f&15.
template <int R>
__device__ __forceinline__ void load_kc(const float* __restrict__ src, int ld, int row0,
                                        int k0, int tid, f32x4 (&reg)[R / 16]) {
#pragma unroll
  for (int i = 0; i < R / 16; ++i) {
    const int f = tid + i * 256;
    const int row = f >> 4, slot = f & 15;
    reg[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + row) * ld + k0 + slot * 4);
  }
}
template <int R>
__device__ __forceinline__ void store_kc(float* lds, int tid, const f32x4 (&reg)[R / 16]) {
#pragma unroll
  for (int i = 0; i < R / 16; ++i) {
    const int f = tid + i * 256;
    const int row = f >> 4, slot = f & 15;
    *reinterpret_cast<f32x4*>(lds + row * 64 + ((slot ^ (row & 15)) << 2)) = reg[i];
  }
}
// KS tile: 64 k-rows x CC cols (stride CC+4).  rows >= kvalid are zero-filled.
template <int CC>
__device__ __forceinline__ void load_ks(const float* __restrict__ src, int ld, int col0,
                                        int k0, int kvalid, int tid, f32x4 (&reg)[CC / 16]) {
  constexpr int C4 = CC / 4;
#pragma unroll
  for (int i = 0; i < CC / 16; ++i) {
    const int f = tid + i * 256;
    const int r = f / C4, c4 = f % C4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < kvalid) v = *reinterpret_cast<const f32x4*>(src + (size_t)(k0 + r) * ld + col0 + c4 * 4);
    reg[i] = v;
  }
}
template <int CC>
__device__ __forceinline__ void store_ks(float* lds, int tid, const f32x4 (&reg)[CC / 16]) {
  constexpr int C4 = CC / 4;
#pragma unroll
  for (int i = 0; i < CC / 16; ++i) {
    const int f = tid + i * 256;
    const int r = f / C4, c4 = f % C4;
    *reinterpret_cast<f32x4*>(lds + r * (CC + 4) + c4 * 4) = reg[i];
  }
}

// ---- the kernel --------------------------------------------------------------

template <int MODE, int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmBatch batch) {
  using Cfg = GemmCfg<MODE, BP, BQ, WP, WQ>;
  constexpr int TP = Cfg::TP, TQ = Cfg::TQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // ---- which problem / tile (XCD-aware: workgroups that share a P panel,
  // i.e. the same weight slice, land on the same XCD's L2; dispatch puts
  // block b on XCD b % 8 — speed only, never correctness)
  int b = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < batch.n && b >= batch.prob[i].tile_base) pi = i;
  const GemmProblem& pr = batch.prob[pi];
  b -= pr.tile_base;
  int tile_p, tile_q;
  if ((pr.tiles_p & 7) == 0) {
    const int xcd = b & 7, j = b >> 3;
    tile_q = j % pr.tiles_q;
    tile_p = (j / pr.tiles_q) * 8 + xcd;
  } else {
    tile_q = b % pr.tiles_q;
    tile_p = b / pr.tiles_q;
  }
  const int p0 = tile_p * BP, q0 = tile_q * BQ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wp = wave % WP, wq = wave / WP;
  const int wp0 = wp * (BP / WP), wq0 = wq * (BQ / WQ);

  const float* __restrict__ Pg = pr.P;
  const float* __restrict__ Qg = pr.Q;
  const int ldp = pr.ldp, ldq = pr.ldq, Kred = pr.Kred;
  const int ntiles = (Kred + 63) >> 6;

  f32x4 acc[TQ][TP];
#pragma unroll
  for (int a = 0; a < TQ; ++a)
#pragma unroll
    for (int c = 0; c < TP; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 rp[Cfg::NP4], rq[Cfg::NQ4];
  float dbsum = 0.0f;  // WGRAD bias gradient (threads < BQ of tile_p == 0 workgroups)

  auto gload = [&](int kt) {
    const int k0 = kt << 6;
    const int kvalid = min(64, Kred - k0);
    if constexpr (Cfg::P_KC) load_kc<BP>(Pg, ldp, p0, k0, tid, rp);
    else load_ks<BP>(Pg, ldp, p0, k0, kvalid, tid, rp);
    if constexpr (Cfg::Q_KC) load_kc<BQ>(Qg, ldq, q0, k0, tid, rq);
    else load_ks<BQ>(Qg, ldq, q0, k0, kvalid, tid, rq);
  };
  auto lstore = [&](int stage) {
    float* ps = smem + stage * Cfg::STAGE_FLOATS;
    float* qs = ps + Cfg::P_FLOATS;
    if constexpr (Cfg::P_KC) store_kc<BP>(ps, tid, rp); else store_ks<BP>(ps, tid, rp);
    if constexpr (Cfg::Q_KC) store_kc<BQ>(qs, tid, rq); else store_ks<BQ>(qs, tid, rq);
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < ntiles) gload(kt + 1);

    const float* ps = smem + stage * Cfg::STAGE_FLOATS;
    const float* qs = ps + Cfg::P_FLOATS;
    const int kvalid = min(64, Kred - (kt << 6));
    const int nkb = (kvalid + 15) >> 4;

    if constexpr (MODE == GEMM_WGRAD) {
      // bias gradient: column sums of the dY tile, m ascending (deterministic)
      if (pr.db != nullptr && tile_p == 0 && tid < BQ) {
        for (int r = 0; r < kvalid; ++r) dbsum += qs[r * (BQ + 4) + tid];
      }
    }

#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb < nkb) {
        float pf[TP][4], qf[TQ][4];
#pragma unroll
        for (int c = 0; c < TP; ++c) {
          if constexpr (Cfg::P_KC) {
            const int row = wp0 + c * 16 + li;
            const f32x4 v = *reinterpret_cast<const f32x4*>(
                ps + row * 64 + ((((kb << 2) + lg) ^ (row & 15)) << 2));
            pf[c][0] = v.x; pf[c][1] = v.y; pf[c][2] = v.z; pf[c][3] = v.w;
          } else {
            const int col = wp0 + c * 16 + li;
#pragma unroll
            for (int s = 0; s < 4; ++s) pf[c][s] = ps[((kb << 4) + (lg << 2) + s) * (BP + 4) + col];
          }
        }
#pragma unroll
        for (int a = 0; a < TQ; ++a) {
          if constexpr (Cfg::Q_KC) {
            const int row = wq0 + a * 16 + li;
            const f32x4 v = *reinterpret_cast<const f32x4*>(
                qs + row * 64 + ((((kb << 2) + lg) ^ (row & 15)) << 2));
            qf[a][0] = v.x; qf[a][1] = v.y; qf[a][2] = v.z; qf[a][3] = v.w;
          } else {
            const int col = wq0 + a * 16 + li;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[a][s] = qs[((kb << 4) + (lg << 2) + s) * (BQ + 4) + col];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < TQ; ++a)
#pragma unroll
            for (int c = 0; c < TP; ++c)
              acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[c][s], qf[a][s], acc[a][c], 0, 0, 0);
      }
    }

    if (kt + 1 < ntiles) lstore(stage ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[q = q0+wq0+a*16+li][p = p0+wp0+c*16+lg*4 .. +3]
  float ssq = 0.0f;
#pragma unroll
  for (int a = 0; a < TQ; ++a) {
    const int q = q0 + wq0 + a * 16 + li;
#pragma unroll
    for (int c = 0; c < TP; ++c) {
      const int p = p0 + wp0 + c * 16 + (lg << 2);
      f32x4 v = acc[a][c];
      if constexpr (MODE == GEMM_FWD) {
        if (pr.bias != nullptr) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(pr.bias + p);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (pr.relu) { v.x = lrelu_fwd(v.x); v.y = lrelu_fwd(v.y); v.z = lrelu_fwd(v.z); v.w = lrelu_fwd(v.w); }
      } else if constexpr (MODE == GEMM_DGRAD) {
        if (pr.mask != nullptr) {
          const f32x4 mv = *reinterpret_cast<const f32x4*>(pr.mask + (size_t)q * pr.ldm + p);
          v.x *= lrelu_mask(mv.x); v.y *= lrelu_mask(mv.y); v.z *= lrelu_mask(mv.z); v.w *= lrelu_mask(mv.w);
        }
      } else {
        ssq = fmaf(v.x, v.x, ssq); ssq = fmaf(v.y, v.y, ssq);
        ssq = fmaf(v.z, v.z, ssq); ssq = fmaf(v.w, v.w, ssq);
      }
      *reinterpret_cast<f32x4*>(pr.C + (size_t)q * pr.ldc + p) = v;
    }
  }

  if constexpr (MODE == GEMM_WGRAD) {
    if (pr.db != nullptr && tile_p == 0 && tid < BQ) {
      pr.db[q0 + tid] = dbsum;
      ssq = fmaf(dbsum, dbsum, ssq);
    }
    if (pr.partial != nullptr) {
      // fixed-shape tree: wave butterfly, then 4 wave sums added in order
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ssq += __shfl_xor(ssq, off, 64);
      __syncthreads();               // all MFMA-phase LDS reads are done
      if (lane == 0) smem[wave] = ssq;
      __syncthreads();
      if (tid == 0) pr.partial[tile_q * pr.tiles_p + tile_p] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    }
  }
}

// ---- host-side launcher --------------------------------------------------------

// Raise the dynamic-LDS limit once per instantiation (not legal inside a stream capture).
template <int MODE, int BP, int BQ, int WP, int WQ>
inline hipError_t gemm_prepare() {
  using Cfg = GemmCfg<MODE, BP, BQ, WP, WQ>;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_kernel<MODE, BP, BQ, WP, WQ>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
}

template <int MODE, int BP, int BQ, int WP, int WQ>
inline hipError_t gemm_launch(GemmBatch& batch, hipStream_t stream) {
  using Cfg = GemmCfg<MODE, BP, BQ, WP, WQ>;
  int base = 0;
  for (int i = 0; i < batch.n; ++i) {
    GemmProblem& p = batch.prob[i];
    p.tiles_p = p.Pdim / BP;
    p.tiles_q = p.Qdim / BQ;
    p.tile_base = base;
    base += p.tiles_p * p.tiles_q;
  }
  batch.total_tiles = base;
  hipLaunchKernelGGL((gemm_mfma_kernel<MODE, BP, BQ, WP, WQ>), dim3(base), dim3(256),
                     Cfg::LDS_BYTES, stream, batch);
  return hipGetLastError();
}

}  // namespace dqnhip
