// gemm_common.hip.h — types shared by the fp32 MFMA GEMM kernels of the learner
// (gemm_direct.hip.h) and the head / optimiser kernels (small_kernels.hip.h).
//
// Every tower GEMM computes C[q][p] = sum_k Pop(p,k) * Qop(q,k) with `p` the contiguous
// output dimension; the three modes replace the Caffe InnerProduct forward/backward GEMMs
// the reference reaches from src/dqn.cpp:751, 904, 923, 963, 1013:
//
//   FWD   Y[m][n]  = lrelu(sum_k X[m][k] W[n][k] + b[n])         P=W  Q=X
//   DGRAD dX[m][j] = (sum_n dY[m][n] W[n][j]) * lrelu'(Xp[m][j])  P=W  Q=dY
//   WGRAD dW[n][j] = sum_m dY[m][n] X[m][j] ; db[n] = sum_m dY[m][n]   P=X  Q=dY
//
// with the ReLU(negative_slope=0.01) forward/backward (src/dqn.cpp:292-301) fused into the
// epilogues.  A launch may carry up to 4 independent problems (grouped GEMM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dqnhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum GemmMode { GEMM_FWD = 0, GEMM_DGRAD = 1, GEMM_WGRAD = 2 };

constexpr float kLeakySlope = 0.01f;  // src/dqn.cpp:300

struct GemmProblem {
  const float* P; int ldp;   // operand that indexes the contiguous output dim
  const float* Q; int ldq;   // operand that indexes the output rows
  float* C; int ldc;         // C[q*ldc + p]
  int Pdim, Qdim, Kred;      // Pdim % BP == 0, Qdim % BQ == 0
  const float* bias;         // FWD: bias[p] (may be null)
  const float* mask; int ldm;// DGRAD: previous activation [q][p] for lrelu' (null: none)
  float* db;                 // WGRAD: bias gradient [q] (null: skip)
  float* partial;            // WGRAD: one sum-of-squares partial per tile (null: skip)
  int relu;                  // FWD: apply leaky ReLU
  // FWD, top tower layer of the critic(s, mu(s)) pass: the epilogue also writes the seed of BackwardFrom(q_values_layer)
  // (src/dqn.cpp:918-923: q diff = -1 per row) taken through the head and this layer's ReLU,
  // C2[q][p] = (-seed_w[p]) * lrelu'(C[q][p]) — what a head-backward launch of its own used to compute from C (null: none)
  const float* seed_w; float* C2;
  // FWD, top tower layer of a critic pass of Step(1): the epilogue also leaves the head's dot product in pieces,
  // dot_out[q][p / 16] = sum over the 16 finished activations C[q][p .. p+15] of C * dot_w (fixed order: a lane's four columns
  // as an fma chain from 0, then (g0 + g1) + (g2 + g3) over the four lane groups) — k_dgrad_qtrain sums the Pdim / 16 pieces of a row
  const float* dot_w; float* dot_out;
  // FWD (first_layers_launch, the state half of critic_target's first layer): the workgroups also leave (column j: row tile j mod tiles_q) columns
  // [xcopy_col, +xcopy_n) of their 16 TP rows of P transposed, xcopy_dst[j][p] = P[p][xcopy_col + j] (the action-column weights the
  // target actor's head kernel applies: there one coalesced float4 per action instead of 40 strided dwords per thread); null: none
  float* xcopy_dst; int xcopy_col, xcopy_n;
  int mode;                  // mixed-mode launches (gemm_bwd_pair_direct): GEMM_DGRAD / GEMM_WGRAD
  int tiles_p, tiles_q, tile_base;
};

constexpr int kMaxGroup = 4;
struct GemmBatch {
  GemmProblem prob[kMaxGroup];
  int n;
  int total_tiles;
};

__device__ __forceinline__ float lrelu_fwd(float x) {
  // Caffe ReLULayer::Forward: max(x,0) + slope*min(x,0)
  return fmaxf(x, 0.0f) + kLeakySlope * fminf(x, 0.0f);
}
__device__ __forceinline__ float lrelu_mask(float y) {
  // Caffe ReLULayer::Backward (in-place: bottom_data is the output y)
  return (y > 0.0f ? 1.0f : 0.0f) + kLeakySlope * (y <= 0.0f ? 1.0f : 0.0f);
}

// Sum over the 64 lanes of a wave with DPP moves (VALU speed) instead of six ds_bpermute butterfly steps: quad
// xor 1, xor 2, half-row mirror, row mirror give every lane its 16-lane row total; row_bcast15 / row_bcast31 chain
// the four row totals into lane 63, which is broadcast.  (Ten heads x six dependent LDS-crossbar shuffles were
// ~2.5 us per row in the one-wave-per-row head kernels.)  Order: fixed, the same in every wave.
__device__ __forceinline__ float wave_sum64(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sticky device flags (DevState::flags, small_kernels.hip.h)
constexpr int kFlagTarget = 1;     // a TD target of the last update(s) was not finite
constexpr int kFlagGradNorm = 2;   // a gradient L2 norm was not finite: that clip+Adam step was skipped

// Data-parallel learners: the per-block loss / q partials reduced into the 4-float tails that ride in the gradient exchange
// ([loss_sum, q_sum, target flag, 0]; tail[2] carries this rank's non-finite-target flag: it is raised from the rank's OWN replay
// shard, so without it one rank would stop with "Target not finite!" while the others walk into the next collective).
// One block of 256 threads, the reduction tree of tick_body (strided partials, butterfly, fixed cross-wave order).  A launch of its
// own (k_tails) or — round 6 — ONE extra block of the net's last backward launch (everything it reads is complete launches earlier).
struct TailsArgs {
  const float* loss_partial; int n_loss; const double* q_partial; int n_q;
  float inv_batch; float* critic_tail; float* actor_tail; const int* flags;
  int on;                          // rider form: 1 = the launch's last block runs tails_block
};
__device__ __forceinline__ void tails_block(const TailsArgs& a, float* sdot /*[4]*/, double* sq /*[4]*/) {
  const int t = threadIdx.x;
  float dot = 0.0f; double qs = 0.0;
  if (a.critic_tail != nullptr) for (int i = t; i < a.n_loss; i += 256) dot += a.loss_partial[i];
  if (a.actor_tail != nullptr) for (int i = t; i < a.n_q; i += 256) qs += a.q_partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off, 64); qs += __shfl_xor(qs, off, 64); }
  if ((t & 63) == 0) { sdot[t >> 6] = dot; sq[t >> 6] = qs; }
  __syncthreads();
  if (t != 0) return;
  if (a.critic_tail != nullptr) {
    dot = (sdot[0] + sdot[1]) + (sdot[2] + sdot[3]);
    a.critic_tail[0] = dot * a.inv_batch / 2.0f;   // EuclideanLoss: dot / num / 2
    a.critic_tail[1] = 0.f; a.critic_tail[2] = (*a.flags & kFlagTarget) ? 1.0f : 0.f; a.critic_tail[3] = 0.f;
  }
  if (a.actor_tail != nullptr) {
    qs = (sq[0] + sq[1]) + (sq[2] + sq[3]);
    a.actor_tail[0] = 0.f; a.actor_tail[1] = (float)qs; a.actor_tail[2] = 0.f; a.actor_tail[3] = 0.f;
  }
}

}  // namespace dqnhip
