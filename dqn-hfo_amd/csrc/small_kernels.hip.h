// small_kernels.hip.h — the HBM/latency-bound kernels around the MFMA GEMMs:
// replay gather/scatter, skinny head layers, TD target + Euclidean loss,
// inverting gradients, fused clip+Adam+soft-update, bookkeeping.
// Each kernel cites the reference lines it replaces (paths under /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "gemm_direct.hip.h"
#include "gemm_common.hip.h"

namespace dqnhip {

constexpr int kNA = 4;    // kActionSize       src/dqn.hpp:20
constexpr int kNP = 6;    // kActionParamSize  src/dqn.hpp:21
constexpr int kNO = 10;   // ActorOutput       src/dqn.hpp:28
constexpr int kAP = 16;   // padded ActorOutput row (64 B)

// Device-resident scalars of one learner (graph-replayable: nothing that
// changes per update is a kernel argument).
struct DevState {
  int ring_head;        // physical slot of logical transition 0
  int ring_size;        // std::deque::size()
  int actor_iter;       // actor_solver_->iter()
  int critic_iter;      // critic_solver_->iter()
  unsigned long long update_counter;  // Philox counter for on-device sampling
  float critic_loss;    // last update's return value .first
  float avg_q;          // .second
  // sticky until dqnhip_read_stats reports and clears them (the reference aborts instead:
  // CHECK(std::isfinite(target)) src/dqn.cpp:898, CHECK(std::isfinite(critic_loss)) :906)
  int flags;            // kFlagTarget | kFlagGradNorm
  int skipped_steps;    // optimiser steps skipped because the gradient norm was not finite
  // Adam's bias correction sqrt(1 - beta2^t) / (1 - beta1^t) of THIS update's actor / critic step, evaluated by a spare
  // block of the update's first launch (k_gather): two double pow() are a ~2 us dependent chain, which every block of
  // k_adam_soft otherwise sits through before its first load (measured: 23.2 -> 21.3 us per launch without it)
  // Two slots: inside a multi-update graph (dqnhip_update_async_n) update u uses slot u & 1, because the gather of
  // update u + 1 — which writes that update's scalars — rides in update u's LAST launch, the optimiser pass that still
  // reads update u's.  Everything else uses slot 0.
  float adam_corr[2][2];   // [slot][actor, critic]
  // ... and the soft-update switch of this update (max_iter() % soft_update_freq == 0 AFTER both increments,
  // src/dqn.cpp:967), from the same block: with both in DevState no block of k_adam_soft reads an iteration counter,
  // so the update's bookkeeping (tick_body) no longer has to wait for the last block of the last launch
  int soft_now[2];
  // multi-update graphs: (update_counter, actor_iter, critic_iter) as the graph's FIRST gather found them.  A gather
  // that rides ahead in the previous update's last launch runs beside the block that advances the live counters, so it
  // takes its own from here: base + its position in the graph (a capture-time constant).
  unsigned long long gbase_counter;
  int gbase_it[2];
};
// (kFlagTarget / kFlagGradNorm: gemm_common.hip.h)

// ---- counter-based RNG (Philox-4x32-10) for SampleTransitionsFromMemory ------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t ctr, uint32_t lane) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), lane, 0x9E3779B9u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  return c[0];
}

// ---- replay ring -------------------------------------------------------------
// SoA ring in HBM: state[cap][SP], next[cap][SP] (rows padded to SP = roundup(S,64)
// floats so every row is whole 256-B lines), act[cap][16], reward[cap], mc[cap],
// term[cap].  Logical index i (what the reference's deque exposes) lives in
// physical slot (head + i) % cap.
struct Ring {
  float* state; float* next; float* act; float* reward; float* mc; uint8_t* term;
  int cap, S, SP;
};

// DQN::AddTransitions / AddTransition (src/dqn.cpp:768-781): the eviction
// arithmetic runs on one thread and publishes (head,size); rows are scattered
// by the rest of the grid from the values BEFORE the update (old_head/old_size
// are recomputed identically by every block).
static __global__ void k_add_transitions(Ring ring, DevState* st, const float* __restrict__ s,
                                  const float* __restrict__ a, const float* __restrict__ r,
                                  const float* __restrict__ mc, const float* __restrict__ nx,
                                  const uint8_t* __restrict__ term, int n, int single_mode,
                                  int* done_counter) {
  // every block derives the same post-eviction (head,size)
  int head = st->ring_head, size = st->ring_size;
  if (single_mode == 2) {      // LoadReplayMemory: plain append, no eviction (caller checked the capacity)
  } else if (single_mode) {    // AddTransition: pop iff size == capacity
    if (size == ring.cap) { head = (head + 1) % ring.cap; size -= 1; }
  } else {                     // AddTransitions: while (size + n >= capacity) pop_front
    int pops = size + n - ring.cap + 1;
    if (pops < 0) pops = 0;
    if (pops > size) pops = size;
    head = (int)(((long long)head + pops) % ring.cap); size -= pops;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row < n) {
    const long long slot = ((long long)head + size + row) % ring.cap;
    const uint8_t t = term[row];
    for (int c = lane; c < ring.SP; c += 64) {
      ring.state[slot * ring.SP + c] = c < ring.S ? s[(size_t)row * ring.S + c] : 0.0f;
      ring.next[slot * ring.SP + c] = (c < ring.S && !t && nx != nullptr) ? nx[(size_t)row * ring.S + c] : 0.0f;
    }
    if (lane < kAP) ring.act[slot * kAP + lane] = lane < kNO ? a[(size_t)row * kNO + lane] : 0.0f;
    if (lane == 0) { ring.reward[slot] = r[row]; ring.mc[slot] = mc[row]; ring.term[slot] = t ? 1 : 0; }
  }
  // last block to finish publishes the new (head,size)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int prev = atomicAdd(done_counter, 1);
    if (prev == (int)gridDim.x - 1) {
      st->ring_head = head; st->ring_size = size + n; *done_counter = 0;
      __threadfence();
    }
  }
}

static __global__ void k_read_memory(Ring ring, const DevState* st, int first, int n, float* s, float* a,
                              float* r, float* mc, float* nx, uint8_t* term) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const long long slot = ((long long)st->ring_head + first + row) % ring.cap;
  for (int c = lane; c < ring.S; c += 64) {
    if (s) s[(size_t)row * ring.S + c] = ring.state[slot * ring.SP + c];
    if (nx) nx[(size_t)row * ring.S + c] = ring.next[slot * ring.SP + c];
  }
  if (a && lane < kNO) a[(size_t)row * kNO + lane] = ring.act[slot * kAP + lane];
  if (lane == 0) {
    if (r) r[row] = ring.reward[slot];
    if (mc) mc[row] = ring.mc[slot];
    if (term) term[row] = ring.term[slot];
  }
}

// DQN::SampleStatesFromMemory (src/dqn.cpp:511-523): one wave per sampled transition, dense [n][S] out
static __global__ void k_sample_states(Ring ring, const DevState* rs, const int* __restrict__ idx_in, uint64_t key,
                                unsigned long long counter, int n, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const int size = rs->ring_size;
  int li = idx_in ? idx_in[row] : (int)(((uint64_t)philox_u32(key, counter, (uint32_t)row) * (uint64_t)size) >> 32);
  li = li < 0 ? 0 : (li >= size ? size - 1 : li);
  const long long slot = ((long long)rs->ring_head + li) % ring.cap;
  for (int c = lane; c < ring.S; c += 64) out[(size_t)row * ring.S + c] = ring.state[slot * ring.SP + c];
}

// Minibatch gather (src/dqn.cpp:846-887): one wave per sampled transition; each
// row of the ring is whole 256-B lines so the reads are fully coalesced.  Writes
// the five network input panels directly (Concat layer, src/dqn.cpp:446-448,
// folded in):  Xa_s=[s|0]  Xa_n=[s'|0]  Xc_tr=[s|a|0]  Xc_pl=[s|0..]  Xc_nx=[s'|0..]
struct GatherOut {
  float* Xa_s; float* Xa_n; int KaP;
  float* Xc_tr; float* Xc_pl; float* Xc_nx; int KcP;
  float* reward; float* mc; float* term; int* idx;
  // fp16 learner: the same five panels as fp16 (what its GEMMs read), written here instead of by a conversion launch
  // (null: fp32 learner).  Row strides = KaP / KcP (the fp16 learner pads both to 128).
  _Float16* Ha_s; _Float16* Ha_n; _Float16* Hc_tr; _Float16* Hc_pl; _Float16* Hc_nx;
};
// rs: the DevState that holds the ring's (head,size) — another learner's under
// ShareReplayMemory; st: this learner's (sampling counter)
// correction = sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in double, rounded once (Caffe's AdamSolver)
__device__ __forceinline__ float adam_correction(float beta1, float beta2, int t) {
  return (float)(sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t)));
}
// One gather: `blocks` - 1 row blocks (4 transitions each) + ONE scalars block (the last) whose first lanes evaluate this
// update's Adam corrections (t = iter + 1 of the actor / the critic: the counters only move in the update's last block)
// and its soft-update switch.
struct GatherArgs {
  Ring ring; const DevState* rs; DevState* st; const int* idx_in; uint64_t seed; GatherOut o; int B;
  float* corr; int* soft_now;       // DevState::adam_corr[slot], &DevState::soft_now[slot]
  float beta1, beta2; int soft_update_freq;
  // -1: a launch of its own — the live counters are this update's.  k >= 1: the gather of the k-th update of a
  // multi-update graph riding in update k-1's last launch — counters = DevState::gbase + k (see DevState).
  // -2: explicit indices, riding in the previous update's CRITIC optimiser launch (a kernel boundary before that update's tick):
  // counters = live + 1.
  int ahead;
  int store_base;                   // 1 (first update of a multi-update graph): also store the live counters to gbase
  int blocks;
};
__device__ __forceinline__ void gather_block(const GatherArgs& g, const int blk) {
  const DevState* st = g.st;
  if (blk == g.blocks - 1) {
    int it_a, it_c;
    if (g.ahead == -2) { it_a = st->actor_iter + 1; it_c = st->critic_iter + 1; }      // rides in the PREVIOUS update's critic launch, before that update's tick (dqnhip_update_chained)
    else if (g.ahead < 0) { it_a = st->actor_iter; it_c = st->critic_iter; }
    else { it_a = st->gbase_it[0] + g.ahead; it_c = st->gbase_it[1] + g.ahead; }
    if (threadIdx.x == 64) *g.soft_now = ((((it_a + 1) > (it_c + 1) ? (it_a + 1) : (it_c + 1)) % g.soft_update_freq) == 0);
    if (threadIdx.x == 65 && g.store_base) { g.st->gbase_counter = st->update_counter; g.st->gbase_it[0] = it_a; g.st->gbase_it[1] = it_c; }
    // four lanes, one pow() each (the two powers of a correction side by side: half the dependent chain), same
    // expression as adam_correction() from there on
    if (threadIdx.x < 64) {
      const int which = (threadIdx.x >> 1) & 1, isb1 = threadIdx.x & 1;
      const int t = (which == 0 ? it_a : it_c) + 1;
      const double pw = pow((double)(isb1 ? g.beta1 : g.beta2), (double)t);
      const double p1 = __shfl_down(pw, 1, 64);          // lane 2*which: pw = beta2^t, p1 = beta1^t
      if (threadIdx.x < 4 && !isb1) g.corr[which] = (float)(sqrt(1.0 - pw) / (1.0 - p1));
    }
    return;
  }
  const GatherOut& o = g.o; const Ring& ring = g.ring;
  const int row = blk * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= g.B) return;
  const int size = g.rs->ring_size;
  int li;
  if (g.idx_in != nullptr) li = g.idx_in[row];
  else {
    // SampleTransitionsFromMemory (src/dqn.cpp:501-509): uniform in [0,size-1] with
    // replacement; counter-based so the draw depends only on (seed, update, row)
    const unsigned long long ctr = g.ahead < 0 ? st->update_counter : st->gbase_counter + (unsigned long long)g.ahead;
    const uint32_t u = philox_u32(g.seed, ctr, (uint32_t)row);
    li = (int)(((uint64_t)u * (uint64_t)size) >> 32);
  }
  li = li < 0 ? 0 : (li >= size ? size - 1 : li);
  const long long slot = ((long long)g.rs->ring_head + li) % ring.cap;
  const float* sp = ring.state + slot * ring.SP;
  const float* np = ring.next + slot * ring.SP;
  const float* ap = ring.act + slot * kAP;
  const int S = ring.S;
  if (o.Xa_s != nullptr)                       // (fp16 learner: nothing reads the fp32 panels — not written)
  for (int c = lane; c < o.KcP; c += 64) {
    const float sv = c < S ? sp[c] : 0.0f;
    const float nv = c < S ? np[c] : 0.0f;
    const float av = (c >= S && c < S + kNO) ? ap[c - S] : 0.0f;
    {
      if (c < o.KaP) { o.Xa_s[(size_t)row * o.KaP + c] = sv; o.Xa_n[(size_t)row * o.KaP + c] = nv; }
      o.Xc_tr[(size_t)row * o.KcP + c] = c < S ? sv : av;
      o.Xc_pl[(size_t)row * o.KcP + c] = sv;
      o.Xc_nx[(size_t)row * o.KcP + c] = nv;
    }
  }
  if (o.Ha_s != nullptr) {
    // fp16 learner: the five panels as fp16, two columns per lane (4-byte stores; 2-byte stores cost ~2x per byte and this
    // kernel writes 5 panels x 256 B per row).  Ring rows are whole 256-B lines (SP = roundup(S, 64) floats), so the pair
    // (c, c + 1) is one 8-byte load wherever c < S.  KaP, KcP are multiples of 128 in fp16 mode.
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    for (int c = lane * 2; c < o.KcP; c += 128) {
      f2 sv = f2{0.f, 0.f}, nv = f2{0.f, 0.f};
      if (c < S) { sv = *reinterpret_cast<const f2*>(sp + c); nv = *reinterpret_cast<const f2*>(np + c); }      // (c even, rows padded to SP >= S + 1: in bounds)
      if (c + 1 >= S) { sv.y = 0.0f; nv.y = 0.0f; }      // whatever the ring holds beyond S never reaches a panel
      const float a0 = (c >= S && c < S + kNO) ? ap[c - S] : 0.0f, a1 = (c + 1 >= S && c + 1 < S + kNO) ? ap[c + 1 - S] : 0.0f;
      const h2 hs = h2{(_Float16)sv.x, (_Float16)sv.y}, hn = h2{(_Float16)nv.x, (_Float16)nv.y};
      if (c < o.KaP) { *reinterpret_cast<h2*>(o.Ha_s + (size_t)row * o.KaP + c) = hs; *reinterpret_cast<h2*>(o.Ha_n + (size_t)row * o.KaP + c) = hn; }
      *reinterpret_cast<h2*>(o.Hc_tr + (size_t)row * o.KcP + c) = h2{(_Float16)(c < S ? sv.x : a0), (_Float16)(c + 1 < S ? sv.y : a1)};
      *reinterpret_cast<h2*>(o.Hc_pl + (size_t)row * o.KcP + c) = hs;
      *reinterpret_cast<h2*>(o.Hc_nx + (size_t)row * o.KcP + c) = hn;
    }
  }
  if (lane == 0) {
    o.reward[row] = ring.reward[slot]; o.mc[row] = ring.mc[slot];
    o.term[row] = ring.term[slot] ? 1.0f : 0.0f; o.idx[row] = li;
  }
}
static __global__ void k_gather(GatherArgs g) { gather_block(g, (int)blockIdx.x); }


// ---- skinny head layers ------------------------------------------------------
// action_layer(4) + actionpara_layer(6) of the actor and q_values_layer(1) of
// the critic (src/dqn.cpp:426-427, 450) are K=H4 dot products per row: one wave
// per row, float4 strips over k, butterfly reduce.
enum HeadMode { HEAD_ACTOR = 0, HEAD_Q = 1, HEAD_Q_TRAIN = 2, HEAD_Q_POLICY = 3 };
// Tower-top reads of the head kernels.  fp32 learner: the fp32 activation panel.  fp16 learner: the fp16 panel the last
// tower layer's GEMM wrote for the next consumer anyway (x16 != null) — every other layer's input is the fp16-rounded
// activation already, and a separate fp32 copy of the tower top cost 16 MB of writes per forward pass + twice the bytes
// in every head kernel at 4096 rows.  The head arithmetic itself stays fp32.
typedef __attribute__((ext_vector_type(4))) _Float16 head_h4;
__device__ __forceinline__ f32x4 head_ld4(const float* x32, const _Float16* x16, size_t idx) {
  if (x16 != nullptr) {
    const head_h4 v = *reinterpret_cast<const head_h4*>(x16 + idx);
    return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
  }
  return *reinterpret_cast<const f32x4*>(x32 + idx);
}
__device__ __forceinline__ float head_ld1(const float* x32, const _Float16* x16, size_t idx) {
  return x16 != nullptr ? (float)x16[idx] : x32[idx];
}
// The same with the panel type fixed at compile time: the hot loops are instantiated once per type and entered through ONE
// branch (HEAD_DISPATCH), so that a per-load pointer test does not sit between the loads of a batch (measured on the fp32
// headline: k_head_q_train 4.9 -> 6.1 us with the test inside the loop).
template <bool IN16> __device__ __forceinline__ f32x4 head_ld4t(const float* x32, const _Float16* x16, size_t idx) {
  if constexpr (IN16) { const head_h4 v = *reinterpret_cast<const head_h4*>(x16 + idx); return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }
  else return *reinterpret_cast<const f32x4*>(x32 + idx);
}
template <bool IN16> __device__ __forceinline__ float head_ld1t(const float* x32, const _Float16* x16, size_t idx) {
  if constexpr (IN16) return (float)x16[idx]; else return x32[idx];
}
#define HEAD_DISPATCH(is16, body) do { if (is16) body(std::true_type{}); else body(std::false_type{}); } while (0)
struct HeadArgs {
  const float* X; int ldx; int H;      // [rows][H] tower top
  const _Float16* X16;                 // fp16 learner: the same panel in fp16 (then X is null)
  const float* W; const float* b;      // [NH][H], [NH]
  int rows;
  // HEAD_ACTOR
  float* out16;                        // [rows][16]
  float* xc; int ldxc; int xc_col;     // also written into a critic input panel (may be null)
  _Float16* xc16; int ldxc16;          // fp16 learner: and into that panel's fp16 copy (may be null)
  // HEAD_ACTOR, the target actor's head inside Step(1) (round 5; null: off): the block that has just formed mu'(s') of a row also
  // FINISHES the first tower layer of critic_target(s', mu'(s')) for that row.  The layer's state half
  // l1_zs[row][n] = sum_{k < S} W1[n][k] s'[k] came out of the update's first GEMM launch (no bias, no ReLU); here
  // l1_y[row][n] = lrelu((l1_zs[row][n] + sum_a W1[n][S + a] mu'[a]) + b1[n]), a in action order (an fma chain on l1_zs).
  // (With the action-column weights read in place — 40 dwords 512 B apart per thread — this kernel took 6.8 instead of 4.9 us.)
  const float* l1_zs; const float* l1_wt; const float* l1_b; float* l1_y; int l1_ld; int l1_n;   // l1_wt[a][n] = W1[n][S + a] (GemmProblem::xcopy_dst); l1_n <= 1024, % 4 == 0
  // HEAD_Q*
  float* q;                            // [rows]
  // HEAD_Q_TRAIN: TD target + Euclidean loss
  const float* q_target; const float* reward; const float* mc; const float* term;
  float* y; float* dq; float* loss_partial;   // loss_partial[gridDim.x]
  double gamma, beta; float inv_batch;
  // HEAD_Q_POLICY
  double* qsum_partial;                // [gridDim.x]
};

struct HeadArgs2 { HeadArgs p[2]; };

// L1: compiled with the target actor's first-layer epilogue (HeadArgs::l1_*; Step(1) only — the acting path and the critic heads
// instantiate L1 = false and carry neither its 44 registers nor its LDS row)
template <int NH, int MODE, bool L1 = false>
__global__ __launch_bounds__(256) void k_head_fwd(HeadArgs2 a2) {
  // one block per row (grid-strided when there are more rows than blocks): the 4 waves split K
  // (each lane one float4 strip per 1024 columns), butterfly within the wave, then the 4 wave
  // sums are added in fixed order.  For H <= 1024 the head weights stay in registers across rows.
  const HeadArgs& a = a2.p[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ float s_acc[4][NH];
  const bool hoist = a.H <= 1024;
  const float bias_j = threadIdx.x < NH ? a.b[threadIdx.x] : 0.0f;      // requested now, used after the reduction
  f32x4 wreg[NH];
  if (hoist) {
#pragma unroll
    for (int j = 0; j < NH; ++j)
      wreg[j] = (threadIdx.x * 4 < a.H) ? *reinterpret_cast<const f32x4*>(a.W + (size_t)j * a.H + threadIdx.x * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // (l1: this thread's four outputs' action-column weights and biases do not depend on the row)
  constexpr bool kL1 = (L1 && MODE == HEAD_ACTOR && NH == kNO);
  __shared__ float s_mu[kAP];
  float l1w[kL1 ? 4 : 1][kL1 ? kNO : 1];
  f32x4 l1b = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool l1_on = kL1 && a.l1_y != nullptr && (int)threadIdx.x * 4 < a.l1_n;
  if constexpr (kL1) {
    if (l1_on) {
#pragma unroll
      for (int j = 0; j < kNO; ++j) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.l1_wt + (size_t)j * a.l1_n + threadIdx.x * 4);
        l1w[0][j] = w4.x; l1w[1][j] = w4.y; l1w[2][j] = w4.z; l1w[3][j] = w4.w;
      }
      l1b = *reinterpret_cast<const f32x4*>(a.l1_b + threadIdx.x * 4);
    }
  }
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
    float acc[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) acc[j] = 0.0f;
    const size_t x0 = (size_t)row * a.ldx;
    f32x4 zs = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (kL1) { if (l1_on) zs = *reinterpret_cast<const f32x4*>(a.l1_zs + (size_t)row * a.l1_ld + threadIdx.x * 4); }
    auto dots = [&](auto tag) {
      for (int k = threadIdx.x * 4; k < a.H; k += 1024) {
        const f32x4 xv = head_ld4t<decltype(tag)::value>(a.X, a.X16, x0 + k);
#pragma unroll
        for (int j = 0; j < NH; ++j) {
          const f32x4 wv = hoist ? wreg[j] : *reinterpret_cast<const f32x4*>(a.W + (size_t)j * a.H + k);
          acc[j] = fmaf(xv.x, wv.x, acc[j]); acc[j] = fmaf(xv.y, wv.y, acc[j]);
          acc[j] = fmaf(xv.z, wv.z, acc[j]); acc[j] = fmaf(xv.w, wv.w, acc[j]);
        }
      }
    };
    HEAD_DISPATCH(a.X16 != nullptr, dots);
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      acc[j] = wave_sum64(acc[j]);
      if (lane == 0) s_acc[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < kAP) {
      const int j = threadIdx.x;
      float v = 0.0f;
      if (j < NH) v = ((s_acc[0][j] + s_acc[1][j]) + (s_acc[2][j] + s_acc[3][j])) + bias_j;
      if constexpr (MODE == HEAD_ACTOR) {
        a.out16[(size_t)row * kAP + j] = v;
        if (a.xc != nullptr && j < NH) a.xc[(size_t)row * a.ldxc + a.xc_col + j] = v;
        if (a.xc16 != nullptr && j < NH) a.xc16[(size_t)row * a.ldxc16 + a.xc_col + j] = (_Float16)v;
        if constexpr (kL1) s_mu[j] = v;
      } else {
        if (j == 0) {
          a.q[row] = v;
          if constexpr (MODE == HEAD_Q_POLICY) a.qsum_partial[row] = (double)v;   // summed in row order by k_tick / k_tails
        }
      }
    }
    __syncthreads();                       // s_acc is rewritten by the next row
    if constexpr (kL1) {
      if (a.l1_y != nullptr) {             // (uniform)
        if (l1_on) {
          float o[4] = {zs.x, zs.y, zs.z, zs.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int j = 0; j < kNO; ++j) o[e] = fmaf(l1w[e][j], s_mu[j], o[e]);
          }
          f32x4 y;
          y.x = lrelu_fwd(o[0] + l1b.x); y.y = lrelu_fwd(o[1] + l1b.y); y.z = lrelu_fwd(o[2] + l1b.z); y.w = lrelu_fwd(o[3] + l1b.w);
          *reinterpret_cast<f32x4*>(a.l1_y + (size_t)row * a.l1_ld + threadIdx.x * 4) = y;
        }
        __syncthreads();                   // s_mu is rewritten by the next row
      }
    }
  }
}

// (Round 4: a form with the head weights staged once per block in LDS, 16 rows per block, all of a wave's rows in flight
// at once was built and measured — 13.4 us against 12.5 for two 4096-row fp16 passes, no change at 2048 fp32 rows: the
// per-wave weight reload is not what this kernel waits for; it streams its panel at ~2.7 TB/s either way.  Not kept.)
// Large minibatches (rows >= 1024): one WAVE per row, no block-level synchronisation; the head weights
// stay in registers across the rows of a wave (H <= 1024: NH x 4 float4 per lane).
template <int NH, int MODE>
__global__ __launch_bounds__(256) void k_head_fwd_rows(HeadArgs2 a2) {
  const HeadArgs& a = a2.p[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // a lane's four float4 strips of a row: fp32 panel k = 4 lane + 256 t (16-B loads); fp16 panel k = 8 lane + 512 (t / 2)
  // + 4 (t % 2), i.e. two 16-B loads of eight halves each (8-B loads run at about half the rate per byte)
  const bool in16 = a.X16 != nullptr;
  auto kof = [&](int t) { return in16 ? lane * 8 + 512 * (t >> 1) + 4 * (t & 1) : lane * 4 + 256 * t; };
  f32x4 wreg[NH][4];
#pragma unroll
  for (int j = 0; j < NH; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = kof(t);
      wreg[j][t] = k < a.H ? *reinterpret_cast<const f32x4*>(a.W + (size_t)j * a.H + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // the next row of this wave is fetched while the current one is reduced (the loop was one exposed memory latency
  // per row: 4 rows per wave at 4096 rows)
  auto load_row = [&](int row, f32x4 (&v)[4]) {
    const size_t x0 = (size_t)row * a.ldx;
    if (in16) {
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
      h8 u[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = lane * 8 + 512 * q;
        if (k < a.H) u[q] = *reinterpret_cast<const h8*>(a.X16 + x0 + k);
        else { for (int e = 0; e < 8; ++e) u[q][e] = (_Float16)0.f; }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        v[2 * q] = f32x4{(float)u[q][0], (float)u[q][1], (float)u[q][2], (float)u[q][3]};
        v[2 * q + 1] = f32x4{(float)u[q][4], (float)u[q][5], (float)u[q][6], (float)u[q][7]};
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = lane * 4 + 256 * t;
        v[t] = k < a.H ? *reinterpret_cast<const f32x4*>(a.X + x0 + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  const int row_step = gridDim.x * 4;
  const float bias_l = lane < NH ? a.b[lane] : 0.0f;   // once: inside the row loop the stores keep it from being hoisted
  f32x4 xn[4];
  if ((int)(blockIdx.x * 4 + wave) < a.rows) load_row(blockIdx.x * 4 + wave, xn);
  for (int row = blockIdx.x * 4 + wave; row < a.rows; row += row_step) {
    f32x4 xv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xv[t] = xn[t];
    if (row + row_step < a.rows) load_row(row + row_step, xn);
    float acc[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s = fmaf(xv[t].x, wreg[j][t].x, s); s = fmaf(xv[t].y, wreg[j][t].y, s);
        s = fmaf(xv[t].z, wreg[j][t].z, s); s = fmaf(xv[t].w, wreg[j][t].w, s);
      }
      acc[j] = wave_sum64(s);
    }
    if (lane < kAP) {
      float v = 0.0f;
#pragma unroll
      for (int j = 0; j < NH; ++j) if (lane == j) v = acc[j] + bias_l;
      if constexpr (MODE == HEAD_ACTOR) {
        a.out16[(size_t)row * kAP + lane] = v;
        if (a.xc != nullptr && lane < NH) a.xc[(size_t)row * a.ldxc + a.xc_col + lane] = v;
        if (a.xc16 != nullptr && lane < NH) a.xc16[(size_t)row * a.ldxc16 + a.xc_col + lane] = (_Float16)v;
      } else {
        if (lane == 0) {
          a.q[row] = v;
          if constexpr (MODE == HEAD_Q_POLICY) a.qsum_partial[row] = (double)v;
        }
      }
    }
  }
}

// Both critic heads of the training step in one launch: q' = q_values(critic_target tower
// top), q = q_values(critic tower top), then the TD target (src/dqn.cpp:892-900, doubles where
// the reference has them) and the EuclideanLoss diff.  One wave per row.
struct HeadTrainArgs {
  const float* Xt; const float* Wt; const float* bt;   // target critic top / head
  const float* X; const float* W; const float* b;      // online critic top / head
  const _Float16* Xt16; const _Float16* X16;           // fp16 learner: the tower tops in fp16 (then Xt / X are null)
  int H, rows;
  const float* reward; const float* mc; const float* term;
  float* q_target; float* q; float* y; float* dq; float* loss_partial;
  double gamma, beta; float inv_batch;
  DevState* st;                                         // non-finite target flag (src/dqn.cpp:898)
  // not null: this launch also writes the online critic's tower-top gradient dZ[row][k] = (dq[row] * W[k]) * lrelu'(X[row][k])
  // — what k_head_bwd<1> computed from dq in a launch of its own.  The wave that forms a row's dq has just streamed that
  // row of X and W through its registers; the head's own dW / db ride elsewhere (head_wgrad_rider), so with this the
  // critic's head-backward launch of Step(1) is gone.
  float* dZ;
  // fp16 learner (round 6): the same gradient as the scaled fp16 panel its GEMMs read, dZ16[row][k] = (h16)(dZ * scale16) — with it
  // the fp16 learner's k_head_bwd<1> / k_head_bwd_big<1> + k_head_wred<1> launches of Step(1) are gone as well
  _Float16* dZ16; float scale16;
  // k_dgrad_qtrain: the two head dot products in 16-column pieces, [rows][H / 16], left by the top forward layers (GemmProblem::dot_w)
  const float* pdt; const float* pd;
};
static __global__ __launch_bounds__(256) void k_head_q_train(HeadTrainArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  __shared__ float s_part[4];
  float at = 0.0f, ao = 0.0f;
  // the row's scalars and the two biases: requested before the dot products, used after the reduction
  float bt0 = 0.0f, b0 = 0.0f, r = 0.0f, mcv = 0.0f, tm = 0.0f;
  if (row < a.rows && lane == 0) { bt0 = a.bt[0]; b0 = a.b[0]; r = a.reward[row]; mcv = a.mc[row]; tm = a.term[row]; }
  if (row < a.rows) {
    const size_t x0 = (size_t)row * a.H;
    auto dots = [&](auto tag) {
      for (int k = lane * 4; k < a.H; k += 256) {
        const f32x4 v0 = head_ld4t<decltype(tag)::value>(a.Xt, a.Xt16, x0 + k), w0 = *reinterpret_cast<const f32x4*>(a.Wt + k);
        const f32x4 v1 = head_ld4t<decltype(tag)::value>(a.X, a.X16, x0 + k), w1 = *reinterpret_cast<const f32x4*>(a.W + k);
        at = fmaf(v0.x, w0.x, at); at = fmaf(v0.y, w0.y, at); at = fmaf(v0.z, w0.z, at); at = fmaf(v0.w, w0.w, at);
        ao = fmaf(v1.x, w1.x, ao); ao = fmaf(v1.y, w1.y, ao); ao = fmaf(v1.z, w1.z, ao); ao = fmaf(v1.w, w1.w, ao);
      }
    };
    HEAD_DISPATCH(a.X16 != nullptr, dots);
  }
  at = wave_sum64(at); ao = wave_sum64(ao);
  float d2 = 0.0f, dq_row = 0.0f;
  if (row < a.rows && lane == 0) {
    const float qt = at + bt0, q = ao + b0;
    a.q_target[row] = qt; a.q[row] = q;
    const float off_policy = tm != 0.0f ? r : (float)((double)r + a.gamma * (double)qt);
    const float target = (float)(a.beta * (double)mcv + (1 - a.beta) * (double)off_policy);
    a.y[row] = target;
    if (!isfinite(target)) atomicOr(&a.st->flags, kFlagTarget);   // CHECK(std::isfinite(target)), src/dqn.cpp:898
    const float d = q - target;
    dq_row = a.inv_batch * d;
    a.dq[row] = dq_row;
    d2 = d * d;
  }
  if ((a.dZ != nullptr || a.dZ16 != nullptr) && row < a.rows) {     // (wave-uniform condition)
    const float dqv = __shfl(dq_row, 0, 64);
    const size_t x0 = (size_t)row * a.H;
    auto seed = [&](auto tag) {
      for (int k = lane * 4; k < a.H; k += 256) {
        const f32x4 xv = head_ld4t<decltype(tag)::value>(a.X, a.X16, x0 + k), wv = *reinterpret_cast<const f32x4*>(a.W + k);
        // k_head_bwd<1>'s arithmetic: s0 = fma(d, w, 0) = d * w, then * lrelu'(x)
        f32x4 dz;
        dz.x = (dqv * wv.x) * lrelu_mask(xv.x); dz.y = (dqv * wv.y) * lrelu_mask(xv.y);
        dz.z = (dqv * wv.z) * lrelu_mask(xv.z); dz.w = (dqv * wv.w) * lrelu_mask(xv.w);
        if constexpr (decltype(tag)::value)
          *reinterpret_cast<head_h4*>(a.dZ16 + x0 + k) = head_h4{(_Float16)(dz.x * a.scale16), (_Float16)(dz.y * a.scale16), (_Float16)(dz.z * a.scale16), (_Float16)(dz.w * a.scale16)};
        else
          *reinterpret_cast<f32x4*>(a.dZ + x0 + k) = dz;
      }
    };
    HEAD_DISPATCH(a.X16 != nullptr, seed);
  }
  if (lane == 0) s_part[wave] = d2;
  __syncthreads();
  if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
}

// ---- Step(1)'s head arithmetic inside the critic's first backward launch (round 5) -------------------------------------------
// k_head_q_train sits between the critics' last forward launch and the critic's top-layer dgrad: a 4.9-us launch-floor link
// whose output the dgrad needs only as a PER-ROW SCALAR.  dZ_L[r][n] = (dq_r w_n) lrelu'(x_rn) = (-dq_r) U[r][n] with
// U = (-w) lrelu'(x), which does not depend on q: the online critic's top forward layer leaves U in its epilogue
// (GemmProblem::seed_w, as the dq = -1 pass already does), the dgrad runs on U, and
//     dZ_{L-1}[r][j] = ((sum_n U[r][n] W[n][j]) * (-dq_r)) * lrelu'(x_{L-1}[r][j])
// takes the scalar in its epilogue.  The two head dot products arrive in 16-column pieces from the critics' top forward layers
// (GemmProblem::dot_w: the finished activations are in that epilogue's registers anyway).  Each of the workgroup's four waves
// requests the pieces of four of the tile's 16 rows once its operand pipeline is primed (2 KB per workgroup), and after its last
// MFMA sums them, forms q', q, the TD target and dq (k_head_q_train's arithmetic from there on), publishes -dq_r in LDS before
// the body's barrier; afterwards the workgroup writes its 64-column slices of dZ_L (the next launch's wgrad reads them).  The
// workgroups of tile column 0 write the per-row outputs.  q', q and dZ_{L-1} differ from the two-launch form by fp32 round-off
// only (the dot product is summed in another fixed order; the scalar is applied after the reduction instead of before):
// DQNHIP_TUNE_SEPARATE_Q_TRAIN restores the two launches.
// Measured on the way (profiles/r05_dgrad_qtrain.txt): a FIFTH wave per workgroup reading the tower tops itself — 17.4 us against
// 8.3 + 4.8 for the two launches (it shares a SIMD and the CU's load path with an MFMA wave); the four MFMA waves reading their
// rows of the tower tops — 12.1 us (the 16 workgroups of a tile row each re-read the same 128 KB: +50% on an L2-bound kernel).
struct QTrainHook {
  const HeadTrainArgs& a; float* s_scale; float* s_dq; int q0; bool owner;
  float v0[4], v1[4];        // rows 4 wave + j: piece `lane` of the target / online head dot product (H / 16 <= 64 pieces)
  float rw[4], mcv[4], tm[4], bt0, b0;
  f32x4 xv, wv;              // this thread's piece of the workgroup's first dZ_L slice (row threadIdx.x / 16, 4 columns)
  int c_first;
  __device__ __forceinline__ QTrainHook(const HeadTrainArgs& a_, float* sc, float* sd, int q0_, int tile_p) : a(a_), s_scale(sc), s_dq(sd), q0(q0_), owner(tile_p == 0), c_first(tile_p * 64) {}
  __device__ __forceinline__ void after_prologue() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = a.H >> 4;
    const int pc = lane < np ? lane : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t x0 = (size_t)(q0 + wave * 4 + j) * np + pc;
      v0[j] = a.pdt[x0]; v1[j] = a.pd[x0];
      const int row = q0 + wave * 4 + j;
      rw[j] = a.reward[row]; mcv[j] = a.mc[row]; tm[j] = a.term[row];
    }
    bt0 = a.bt[0]; b0 = a.b[0];
    if (c_first < a.H) {
      const int rr = threadIdx.x >> 4, c = c_first + ((threadIdx.x & 15) << 2);
      xv = *reinterpret_cast<const f32x4*>(a.X + (size_t)(q0 + rr) * a.H + c); wv = *reinterpret_cast<const f32x4*>(a.W + c);
    }
  }
  __device__ __forceinline__ void before_park() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = lane < (a.H >> 4);
    float d2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = wave * 4 + j, row = q0 + rr;
      const float at = wave_sum64(in ? v0[j] : 0.0f), ao = wave_sum64(in ? v1[j] : 0.0f);
      const float r = rw[j];
      const float qt = at + bt0, q = ao + b0;
      const float off_policy = tm[j] != 0.0f ? r : (float)((double)r + a.gamma * (double)qt);      // src/dqn.cpp:893-900
      const float target = (float)(a.beta * (double)mcv[j] + (1 - a.beta) * (double)off_policy);
      const float d = q - target;
      const float dq_row = a.inv_batch * d;
      d2[j] = d * d;
      if (lane == 0) {
        s_scale[rr] = -dq_row; s_dq[rr] = dq_row;
        if (owner) {
          a.q_target[row] = qt; a.q[row] = q; a.y[row] = target; a.dq[row] = dq_row;
          if (!isfinite(target)) atomicOr(&a.st->flags, kFlagTarget);   // CHECK(std::isfinite(target)), src/dqn.cpp:898
        }
      }
    }
    // k_head_q_train's partial of its block of four rows (one row per wave there: ((w0 + w1) + w2) + w3)
    if (owner && lane == 0) a.loss_partial[(q0 >> 2) + wave] = ((d2[0] + d2[1]) + d2[2]) + d2[3];
  }
};
template <int UNUSED = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dgrad_qtrain(const GemmBatch batch, const HeadTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_scale[16], s_dq[16];
  int pi, tile_p, tile_q;
  tile_of_block(batch, pi, tile_p, tile_q);
  const GemmProblem& pr = batch.prob[0];
  const int q0 = tile_q * 16;
  QTrainHook hook(a, s_scale, s_dq, q0, tile_p);
  dgrad_lds_body<1, 1, true, QTrainHook>(pr, tile_p, tile_q, smem, s_scale, hook);
  // this workgroup's slices of dZ_L: columns [64 tile_p, +64) and every tiles_p-th slice after it, 16 rows (one float4 per thread);
  // s_dq was published before the body's barrier
  for (int c0 = tile_p * 64; c0 < a.H; c0 += pr.tiles_p * 64) {
    const int rr = threadIdx.x >> 4, c = c0 + ((threadIdx.x & 15) << 2);
    const size_t x0 = (size_t)(q0 + rr) * a.H + c;
    f32x4 xv = hook.xv, wv = hook.wv;       // (the first slice's operands were requested under the main loop)
    if (c0 != tile_p * 64) { xv = *reinterpret_cast<const f32x4*>(a.X + x0); wv = *reinterpret_cast<const f32x4*>(a.W + c); }
    const float dqv = s_dq[rr];
    f32x4 dz;
    dz.x = (dqv * wv.x) * lrelu_mask(xv.x); dz.y = (dqv * wv.y) * lrelu_mask(xv.y);
    dz.z = (dqv * wv.z) * lrelu_mask(xv.z); dz.w = (dqv * wv.w) * lrelu_mask(xv.w);
    *reinterpret_cast<f32x4*>(a.dZ + x0) = dz;
  }
}
inline hipError_t dgrad_qtrain_launch(GemmBatch& batch, const HeadTrainArgs& a, hipStream_t stream) {
  GemmProblem& p = batch.prob[0];
  p.tiles_p = p.Pdim / 64; p.tiles_q = p.Qdim / 16; p.tile_base = 0;
  batch.n = 1; batch.total_tiles = p.tiles_p * p.tiles_q;
  LaunchTimer& lt = launch_timer();
  if (lt.start) { hipExtLaunchKernelGGL(k_dgrad_qtrain<0>, dim3(batch.total_tiles), dim3(256), (dgrad_lds_bytes<1, 1>()), stream, lt.start, lt.stop, 0, batch, a); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL(k_dgrad_qtrain<0>, dim3(batch.total_tiles), dim3(256), (dgrad_lds_bytes<1, 1>()), stream, batch, a);
  return hipGetLastError();
}

// Fused head backward: in one pass over the tower top X4[rows][H]
//   (actor only) inverting gradients (src/dqn.cpp:924-957) on the critic's input-gradient
//                columns -> dYh[m][0..9]
//   dZ4[m][k]  = (sum_j dYh[m][j] Wh[j][k]) * lrelu'(X4[m][k])       (head dgrad + ReLU bwd)
//   dWh[j][k]  = sum_m dYh[m][j] X4[m][k] ;  dbh[j] = sum_m dYh[m][j]  (head wgrad)
//   + one sum-of-squares partial per block.
// Block = 64 columns x 16 row groups (1024 threads); row groups are added in fixed order.
// Replaces three launches (invert, head dgrad, head wgrad) and 16x the parallelism of the
// old column-strip wgrad.
struct HeadBwdArgs {
  const float* dyh; int lddy;          // NH==1: dq[rows] (null: -1 per row, no wgrad)
  const float* dXc; int ldx; int S;    // actor: critic input gradient (invert source)
  const float* aout16;                 // actor: mu(s) for the inverting bounds
  float* dA16;                         // actor: post-invert head diffs (debug / parity)
  const float* W; const float* X4; int H; int rows;
  const _Float16* X416;                // fp16 learner: the tower top in fp16 (then X4 is null)
  float* dZ; float* dW; float* db; float* partial;
  float* slab;                         // [RC][H/64][NH][64] per-row-chunk partial dW
  int* ticket;                         // [H/64] arrival counters, zero before and after every launch
  // rider (NH == 1, dq = -1 pass): the y-rows >= rc_blocks of the grid compute q = head(X4) + avg-Q partials
  // — the critic(s, mu(s)) head forward (src/dqn.cpp:913-916).  The -1 seed does not depend on q, so the
  // two used to be separate dependent launches for no reason.
  int rc_blocks;                       // row chunks of the backward part (0: gridDim.y)
  const float* q_bias; float* q_out; double* qsum_partial;
  // the rider's own head (fp16 learner: the critic(s, mu(s)) head rides in the ACTOR heads' backward launch — its seed comes
  // out of the critic's top forward layer, HGemm::seed_w); null: the launch's own W / X4 / X416 / H (the dq = -1 launch)
  const float* qr_W; const float* qr_X4; const _Float16* qr_X416; int qr_H;
  // fp16 learner: also emit the tower-top gradient as the scaled fp16 panel the fp16 GEMMs read (dZ16 [rows][H])
  // instead of a separate conversion launch
  _Float16* dZ16; float scale16;
};
// Grid = (H/64 column blocks) x (RC row chunks); block = 64 columns x 16 row groups.  Each block
// writes its dZ rows directly and a partial dW slab; the LAST block to arrive for a column block
// (agent-scope release -> ticket -> acquire, guide G16) adds the RC slabs in fixed order, so the
// result does not depend on which block that is.
// cross-block hand-off words: coherent at agent scope without a fence on either side
__device__ __forceinline__ void slab_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float slab_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int NH>
__global__ __launch_bounds__(1024) void k_head_bwd(HeadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int RC = a.rc_blocks > 0 ? a.rc_blocks : gridDim.y, rc = blockIdx.y, nkb = gridDim.x;
  if (a.q_out != nullptr && rc >= RC) {                // rider blocks: one wave per row, k-strips of float4
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = (((int)blockIdx.y - RC) * (int)gridDim.x + (int)blockIdx.x) * 16 + wave;
    if (row >= a.rows) return;
    const float* rW = a.qr_W ? a.qr_W : a.W; const float* rX = a.qr_W ? a.qr_X4 : a.X4;
    const _Float16* rX16 = a.qr_W ? a.qr_X416 : a.X416; const int rH = a.qr_W ? a.qr_H : a.H;
    const size_t x0 = (size_t)row * rH;
    float acc = 0.0f;
    auto dots = [&](auto tag) {
      for (int k = lane * 4; k < rH; k += 256) {
        const f32x4 xv = head_ld4t<decltype(tag)::value>(rX, rX16, x0 + k), wv = *reinterpret_cast<const f32x4*>(rW + k);
        acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
      }
    };
    HEAD_DISPATCH(rX16 != nullptr, dots);
    acc = wave_sum64(acc);
    if (lane == 0) { const float v = acc + a.q_bias[0]; a.q_out[row] = v; a.qsum_partial[row] = (double)v; }
    return;
  }
  const int rows_c = (a.rows + RC - 1) / RC;           // rows of this chunk
  const int r0 = rc * rows_c, r1 = min(a.rows, r0 + rows_c);
  float* s_dy = sm;                                    // [rows_c][NH]
  float* s_acc = sm + rows_c * NH;                     // [16][NH][64]
  __shared__ int s_last;
  const int tid = threadIdx.x;
  const bool want_w = a.dW != nullptr;
  const int kc = tid & 63, rg = tid >> 6;              // 16 row groups
  const int k = blockIdx.x * 64 + kc;
  const int per = (r1 - r0 + 15) / 16;
  const int m0 = r0 + rg * per, m1 = min(r1, m0 + per);
  // this thread's head weights and its first four tower-top values go out BEFORE the head diffs are staged: neither
  // depends on them, and the staging (two dependent loads + the inverting-gradients arithmetic + a barrier) is a chain
  // of its own.  Rows are then taken four at a time, four independent loads in flight, instead of one load per iteration.
  constexpr int RB = 4;
  float w[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) w[j] = a.W[(size_t)j * a.H + k];
  float xpre[RB];
  auto ldx = [&](int m) -> float { return a.X416 != nullptr ? (float)a.X416[(size_t)m * a.H + k] : a.X4[(size_t)m * a.H + k]; };
#pragma unroll
  for (int u = 0; u < RB; ++u) xpre[u] = (m0 + u < m1) ? ldx(m0 + u) : 0.0f;
  // ---- head diffs of this chunk's rows into LDS
  for (int i = tid; i < (r1 - r0) * NH; i += 1024) {
    const int m = r0 + i / NH, j = i % NH;
    float d;
    if constexpr (NH == kNO) {
      d = a.dXc[(size_t)m * a.ldx + a.S + j];
      const float out = a.aout16[(size_t)m * kAP + j];
      float mn, mx;
      if (j < kNA) { mn = -1.0f; mx = 1.0f; }
      else { const int p = j - kNA; if (p == 0 || p == 4) { mn = 0.0f; mx = 100.0f; } else { mn = -180.0f; mx = 180.0f; } }
      if (d < 0) d *= (mx - out) / (mx - mn);
      else if (d > 0) d *= (out - mn) / (mx - mn);
      if (blockIdx.x == 0) a.dA16[(size_t)m * kAP + j] = d;
    } else {
      d = a.dyh ? a.dyh[(size_t)m * a.lddy + j] : -1.0f;
    }
    s_dy[i] = d;
  }
  __syncthreads();
  float acc[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) acc[j] = 0.0f;
  auto rows_loop = [&](auto tag) {
  for (int mb = m0; mb < m1; mb += RB) {
    float xb[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u)
      xb[u] = (mb == m0) ? xpre[u] : ((mb + u < m1) ? head_ld1t<decltype(tag)::value>(a.X4, a.X416, (size_t)(mb + u) * a.H + k) : 0.0f);
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int m = mb + u;
      if (m >= m1) break;
      const float xv = xb[u];
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const float d = s_dy[(m - r0) * NH + j];
        // Split layer: action_layer's and actionpara_layer's bottom diffs are
        // formed separately and added
        if (NH == kNO && j >= kNA) s1 = fmaf(d, w[j], s1); else s0 = fmaf(d, w[j], s0);
        acc[j] = fmaf(d, xv, acc[j]);
      }
      if (NH == kNO) s0 += s1;
      const float dz = s0 * lrelu_mask(xv);
      if (a.dZ != nullptr) a.dZ[(size_t)m * a.H + k] = dz;
      if (a.dZ16 != nullptr) a.dZ16[(size_t)m * a.H + k] = (_Float16)(dz * a.scale16);
    }
  }
  };
  HEAD_DISPATCH(a.X416 != nullptr, rows_loop);
  if (!want_w) return;
#pragma unroll
  for (int j = 0; j < NH; ++j) s_acc[(rg * NH + j) * 64 + kc] = acc[j];
  __syncthreads();
  // this chunk's partial: row groups added in fixed order
  float* my_slab = a.slab + ((size_t)rc * nkb + blockIdx.x) * NH * 64;
  if (rg < NH) {                                      // wave j folds head j (NH <= 16 waves)
    const int j = rg;
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += s_acc[(g * NH + j) * 64 + kc];
    slab_st(&my_slab[j * 64 + kc], v);
  }
  if (rg == NH && blockIdx.x == 0 && kc < NH) {       // partial bias gradient of this chunk (a spare wave)
    float v = 0.0f;
    for (int m = 0; m < r1 - r0; ++m) v += s_dy[m * NH + kc];
    slab_st(&a.slab[(size_t)RC * nkb * NH * 64 + rc * 16 + kc], v);
  }
  // publish the slab, take a ticket; the last arriver reduces (placement independent).  The slab travels as
  // agent-scope (sc1: write-through / L2-bypassing) stores and loads, drained before the ticket — not plain stores +
  // a release fence (the fence writes back the whole L2, this launch's dZ rows included, once per block, and the
  // last arriver's acquire invalidates it again).  Measured alternatives at 256 rows, same-box A/B: one row chunk on
  // 32-column blocks (no counter): no gain; the gradients from an extra grid row of blocks that walk all rows: -0.8 %.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // The hand-off is MI355X_MICROARCH.md's "handoff-flag" form for gfx950: write-through (sc1) payload stores, the storing
    // waves' `s_waitcnt vmcnt(0)` above, a barrier, then a relaxed agent-scope counter; the last arriver reads the slabs with
    // sc1 (L1-bypassing) loads.  No release / acquire fence: an agent-scope release writes the whole L2 back — this launch's
    // dZ rows included — once per block (measured round 4 with ACQ_REL here: k_head_bwd<10> 9.6 -> 10.7 us, <1> 6.2 -> 6.8
    // at 512 fp16 rows).  That is an ISA-level guarantee, not one of the HIP memory model, so it is pinned by
    // a stress test that checks every word of the reduced gradient under uneven load from a second stream
    // (tests/test_gpu_head_handoff.py); shapes whose head gradients ride in a carrier launch never come here.
    const int t = __hip_atomic_fetch_add(a.ticket + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == RC - 1);
  }
  __syncthreads();
  if (!s_last) return;
  float ssq = 0.0f;
  if (rg < NH) {
    const int j = rg;
    float v = 0.0f;
    for (int c = 0; c < RC; ++c) v += slab_ld(&a.slab[((size_t)c * nkb + blockIdx.x) * NH * 64 + j * 64 + kc]);
    a.dW[(size_t)j * a.H + k] = v;
    ssq = v * v;
  } else if (rg == NH && blockIdx.x == 0 && kc < NH) {
    float v = 0.0f;
    for (int c = 0; c < RC; ++c) v += slab_ld(&a.slab[(size_t)RC * nkb * NH * 64 + c * 16 + kc]);
    a.db[kc] = v;
    ssq = v * v;
  }
  ssq = wave_sum64(ssq);
  if (kc == 0) s_acc[rg] = ssq;                        // s_acc is free again
  __syncthreads();
  if (tid == 0) {
    float t = 0.0f;
#pragma unroll
    for (int g = 0; g <= NH; ++g) t += s_acc[g];        // heads in order, then the bias wave
    if (a.partial != nullptr) a.partial[blockIdx.x] = t;
    a.ticket[blockIdx.x] = 0;                          // re-arm for the next launch
  }
}

// ---- critic dQ/da + inverting gradients + actor heads' backward in ONE launch (round 5) ----------------------------------------
// The critic's BackwardFrom(q_values) chain (src/dqn.cpp:918-923) ends in the first layer's input gradient, of which only the ten
// action columns are consumed; the inverting-gradients pass (:924-957) and the actor heads' backward (BackwardFrom(actionpara_layer),
// :960-963: head dgrad + the tower top's ReLU') follow.  They used to be two launch-floor launches (gemm_dgrad_narrow_qrider:
// 32 tiles of 16 x 16 + the q riders, 6.2 us; k_head_bwd<10>: 4.8 us).  Here workgroup (row tile t, column chunk c) recomputes
// the 16 x 16 tile of dQ/da for its 16 rows — a 1024-deep reduction on four waves, 0.1 us of MFMA: recomputing it in each of the
// H / 256 column chunks costs less than handing it across a kernel boundary — inverts it, and writes its 16 x 256 piece of the
// actor's tower-top gradient.  Same arithmetic in the same order as the two kernels it replaces (bit-identical: the narrow
// tile's values do not depend on which columns a workgroup owns, the head part is k_head_bwd's loop).  The q(s, mu(s)) riders
// come last in the grid, as before.  The head's own dW / db are the wgrad-tail launch's riders (HeadWgradRider).
struct DqdaHeadArgs {
  GemmProblem pr;                          // critic first layer's narrow dgrad: P = W_0 + S (16 columns from the first action column), Q = dZ_1, Kred = width of layer 1
  const float* aout16; float* dA16;        // mu(s) [rows][16]; post-invert diffs [rows][16] (column chunk 0 writes them)
  const float* W; const float* X4;         // actor head weights [10][H], actor tower top [rows][H]
  float* dZ;                               // actor tower-top gradient [rows][H]
  int H, rows, row_tiles;                  // row_tiles = rows / 16
  // fp16 learner (F16 = true, round 6: its layer-0 dgrad launch + k_head_bwd<10> in one): the tile from the fp16 operands
  // (dgrad_narrow_tile16; t16.P = W16_0 + S, t16.Q = the scaled dZ16_1), times inv_ls; the tower top read as fp16 (X416), the
  // tower-top gradient written as the scaled fp16 panel dZ16 = (h16)(dZ * scale16)
  NarrowTile16 t16; float inv_ls;
  const _Float16* X416; _Float16* dZ16; float scale16;
};
template <bool F16 = false>
__global__ __launch_bounds__(256) void k_dqda_head_bwd(const DqdaHeadArgs a, const QHeadRider rider) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // the narrow dgrad's parking area (4 waves x 64 lanes x 16 B)
  __shared__ float s_d[16][17];
  __shared__ float s_dy[16 * kNO];
  const int tiles = a.row_tiles * ((a.H + 255) >> 8);
  if ((int)blockIdx.x >= tiles) { q_head_rider(rider, (int)blockIdx.x - tiles); return; }
  const int rt = (int)blockIdx.x % a.row_tiles, cc = (int)blockIdx.x / a.row_tiles;
  const int tid = threadIdx.x, q0 = rt * 16;
  // (round 6: any tower-top width — the reference's own tower ends in 128 units; threads beyond it keep column H - 1's loads, take
  // part in the tile and the barriers, and store nothing)
  const bool live = (cc << 8) + tid < a.H;
  const int k = live ? (cc << 8) + tid : a.H - 1;
  // everything the head part needs that does not depend on dQ/da goes out first: this thread's ten head weights, its column of
  // the 16 tower-top rows, and (160 threads) one output of mu(s)
  float wh[kNO], xv[16];
#pragma unroll
  for (int j = 0; j < kNO; ++j) wh[j] = a.W[(size_t)j * a.H + k];
#pragma unroll
  for (int r = 0; r < 16; ++r) xv[r] = F16 ? (float)a.X416[(size_t)(q0 + r) * a.H + k] : a.X4[(size_t)(q0 + r) * a.H + k];
  float out = 0.0f;
  if (tid < 16 * kNO) out = a.aout16[(size_t)(q0 + tid / kNO) * kAP + tid % kNO];
  f32x4 v;
  if constexpr (F16) { v = dgrad_narrow_tile16<8>(a.t16, rt, smem); v.x *= a.inv_ls; v.y *= a.inv_ls; v.z *= a.inv_ls; v.w *= a.inv_ls; }
  else v = dgrad_narrow_tile<8>(a.pr, 0, rt, smem);
  if (tid < 64) {                          // wave 0 holds the tile: lane (li, lg), register r = dX[row q0 + li][column 4 lg + r]
    const int li = tid & 15, lg = tid >> 4;
    s_d[li][(lg << 2) + 0] = v.x; s_d[li][(lg << 2) + 1] = v.y; s_d[li][(lg << 2) + 2] = v.z; s_d[li][(lg << 2) + 3] = v.w;
  }
  __syncthreads();
  if (tid < 16 * kNO) {                    // inverting gradients (k_head_bwd<10>'s staging loop)
    const int r = tid / kNO, j = tid % kNO;
    float d = s_d[r][j];
    float mn, mx;
    if (j < kNA) { mn = -1.0f; mx = 1.0f; }
    else { const int p = j - kNA; if (p == 0 || p == 4) { mn = 0.0f; mx = 100.0f; } else { mn = -180.0f; mx = 180.0f; } }
    if (d < 0) d *= (mx - out) / (mx - mn);
    else if (d > 0) d *= (out - mn) / (mx - mn);
    if (cc == 0) a.dA16[(size_t)(q0 + r) * kAP + j] = d;
    s_dy[tid] = d;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int j = 0; j < kNO; ++j) {
      const float d = s_dy[r * kNO + j];
      // Split layer: action_layer's and actionpara_layer's bottom diffs are formed separately and added
      if (j >= kNA) s1 = fmaf(d, wh[j], s1); else s0 = fmaf(d, wh[j], s0);
    }
    s0 += s1;
    const float dz = s0 * lrelu_mask(xv[r]);
    if (!live) continue;
    if constexpr (F16) a.dZ16[(size_t)(q0 + r) * a.H + k] = (_Float16)(dz * a.scale16);
    else a.dZ[(size_t)(q0 + r) * a.H + k] = dz;
  }
}
inline hipError_t dqda_head_bwd_launch(DqdaHeadArgs& a, const QHeadRider& rider, hipStream_t stream) {
  a.row_tiles = a.rows / 16;
  a.pr.tiles_p = 1; a.pr.tiles_q = a.row_tiles; a.pr.tile_base = 0;
  const int grid = a.row_tiles * ((a.H + 255) / 256) + rider.blocks;
  LaunchTimer& lt = launch_timer();
  if (a.dZ16 != nullptr) {
    if (lt.start) { hipExtLaunchKernelGGL(k_dqda_head_bwd<true>, dim3(grid), dim3(256), 4 * 64 * 16, stream, lt.start, lt.stop, 0, a, rider); lt.start = lt.stop = nullptr; }
    else hipLaunchKernelGGL(k_dqda_head_bwd<true>, dim3(grid), dim3(256), 4 * 64 * 16, stream, a, rider);
  }
  else if (lt.start) { hipExtLaunchKernelGGL(k_dqda_head_bwd<false>, dim3(grid), dim3(256), 4 * 64 * 16, stream, lt.start, lt.stop, 0, a, rider); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL(k_dqda_head_bwd<false>, dim3(grid), dim3(256), 4 * 64 * 16, stream, a, rider);
  return hipGetLastError();
}

// ---- head backward for large minibatches (rows >= 1024) -------------------------------------
// Same arithmetic as k_head_bwd, re-tiled for bandwidth: block = 64 rows x 256 columns, wave = 16
// rows, lane = 4 consecutive columns (16-B loads / stores of X4 and dZ).  Per-chunk partial head
// gradients go to a slab [rows/64][NH][H]; k_head_wred adds the chunks in index order.  In fp16
// mode the tower-top gradient is written directly as the scaled fp16 panel, replacing the fp32 panel +
// conversion pass.
struct HeadBwdBigArgs {
  HeadBwdArgs a;
  _Float16* dZ16; float scale16;                             // fp16 output (null: fp32 a.dZ only)
  float* slab2;                                              // [rows/64][NH][H] then [rows/64][16]
  // rider (NH == 1, dq = -1 pass; a.q_out != null): blocks with blockIdx.x >= chunks compute q = head(X4) + the avg-Q
  // partials (critic(s, mu(s)) head forward, src/dqn.cpp:913-916), one wave per row — as in k_head_bwd
  int chunks;
};
template <int NH>
__global__ __launch_bounds__(256) void k_head_bwd_big(HeadBwdBigArgs b) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const HeadBwdArgs& a = b.a;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_dy = sm;                                    // [64][NH]
  float* s_red = sm + 64 * NH;                         // [4][NH][256]
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  if (b.chunks > 0 && (int)blockIdx.x >= b.chunks) {
    // (rider: the q head of HeadBwdArgs::qr_* when given — the fp16 learner's critic(s, mu(s)) head in the ACTOR heads' launch —
    // else this launch's own head, the dq = -1 launch)
    const float* rW = a.qr_W ? a.qr_W : a.W; const float* rX4 = a.qr_W ? a.qr_X4 : a.X4;
    const _Float16* rX416 = a.qr_W ? a.qr_X416 : a.X416; const int rH = a.qr_W ? a.qr_H : a.H;
    // one wave per row, four rows of the wave in flight at once (16 x 16-B loads per lane before the first use:
    // with one row at a time the rider was a chain of exposed memory latencies, 9 us at 4096 rows)
    const int nwave = ((int)gridDim.x - b.chunks) * (int)gridDim.y * 4;
    const int wv = (((int)blockIdx.x - b.chunks) * (int)gridDim.y + (int)blockIdx.y) * 4 + w;
    const bool hoist = rH <= 1024;
    f32x4 wreg[4];
    if (hoist) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { const int k = lane * 4 + 256 * t; wreg[t] = k < rH ? *reinterpret_cast<const f32x4*>(rW + k) : f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    for (int r0 = wv * 4; r0 < a.rows; r0 += nwave * 4) {
      if (hoist) {
        f32x4 xr[4][4];
        if (rX416 != nullptr) {
          head_h4 hr[4][4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int k = lane * 4 + 256 * t;
              hr[j][t] = (r0 + j < a.rows && k < rH) ? *reinterpret_cast<const head_h4*>(rX416 + (size_t)(r0 + j) * rH + k) : head_h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) xr[j][t] = f32x4{(float)hr[j][t].x, (float)hr[j][t].y, (float)hr[j][t].z, (float)hr[j][t].w};
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int k = lane * 4 + 256 * t;
              xr[j][t] = (r0 + j < a.rows && k < rH) ? *reinterpret_cast<const f32x4*>(rX4 + (size_t)(r0 + j) * rH + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float acc = 0.0f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc = fmaf(xr[j][t].x, wreg[t].x, acc); acc = fmaf(xr[j][t].y, wreg[t].y, acc);
            acc = fmaf(xr[j][t].z, wreg[t].z, acc); acc = fmaf(xr[j][t].w, wreg[t].w, acc);
          }
          acc = wave_sum64(acc);
          if (lane == 0 && r0 + j < a.rows) { const float v = acc + a.q_bias[0]; a.q_out[r0 + j] = v; a.qsum_partial[r0 + j] = (double)v; }
        }
      } else {
        for (int j = 0; j < 4 && r0 + j < a.rows; ++j) {
          const size_t x0 = (size_t)(r0 + j) * rH;
          float acc = 0.0f;
          for (int k = lane * 4; k < rH; k += 256) {
            const f32x4 xv = head_ld4(rX4, rX416, x0 + k), wv4 = *reinterpret_cast<const f32x4*>(rW + k);
            acc = fmaf(xv.x, wv4.x, acc); acc = fmaf(xv.y, wv4.y, acc); acc = fmaf(xv.z, wv4.z, acc); acc = fmaf(xv.w, wv4.w, acc);
          }
          acc = wave_sum64(acc);
          if (lane == 0) { const float v = acc + a.q_bias[0]; a.q_out[r0 + j] = v; a.qsum_partial[r0 + j] = (double)v; }
        }
      }
    }
    return;
  }
  const int m0 = blockIdx.x * 64, kb = blockIdx.y * 256, k0 = kb + lane * 4;
  const bool want_w = a.dW != nullptr;
  // the head weights and all 16 rows of this wave's strip of the tower top go out BEFORE the head diffs are staged: neither
  // depends on them, and the staging (two dependent loads, the inverting-gradients arithmetic, a barrier) is a memory round
  // trip of its own that used to sit in front of these loads (round 4: one exposed latency less per launch)
  f32x4 wv[NH], acc[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) { wv[j] = *reinterpret_cast<const f32x4*>(a.W + (size_t)j * a.H + k0); acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 xr[16];
  head_h4 hr[16];
  if (a.X416 != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) hr[r] = *reinterpret_cast<const head_h4*>(a.X416 + (size_t)(m0 + w * 16 + r) * a.H + k0);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[r] = *reinterpret_cast<const f32x4*>(a.X4 + (size_t)(m0 + w * 16 + r) * a.H + k0);
  }
  for (int i = tid; i < 64 * NH; i += 256) {
    const int m = m0 + i / NH, j = i % NH;
    float d;
    if constexpr (NH == kNO) {
      d = a.dXc[(size_t)m * a.ldx + a.S + j];
      const float out = a.aout16[(size_t)m * kAP + j];
      float mn, mx;
      if (j < kNA) { mn = -1.0f; mx = 1.0f; }
      else { const int p = j - kNA; if (p == 0 || p == 4) { mn = 0.0f; mx = 100.0f; } else { mn = -180.0f; mx = 180.0f; } }
      if (d < 0) d *= (mx - out) / (mx - mn);
      else if (d > 0) d *= (out - mn) / (mx - mn);
      if (blockIdx.y == 0) a.dA16[(size_t)m * kAP + j] = d;
    } else {
      d = a.dyh ? a.dyh[(size_t)m * a.lddy + j] : -1.0f;
    }
    s_dy[i] = d;
  }
  __syncthreads();
  if (a.X416 != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[r] = f32x4{(float)hr[r].x, (float)hr[r].y, (float)hr[r].z, (float)hr[r].w};
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ml = w * 16 + r, m = m0 + ml;
    const f32x4 x = xr[r];
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const float d = s_dy[ml * NH + j];
      if (NH == kNO && j >= kNA) { s1.x = fmaf(d, wv[j].x, s1.x); s1.y = fmaf(d, wv[j].y, s1.y); s1.z = fmaf(d, wv[j].z, s1.z); s1.w = fmaf(d, wv[j].w, s1.w); }
      else { s0.x = fmaf(d, wv[j].x, s0.x); s0.y = fmaf(d, wv[j].y, s0.y); s0.z = fmaf(d, wv[j].z, s0.z); s0.w = fmaf(d, wv[j].w, s0.w); }
      acc[j].x = fmaf(d, x.x, acc[j].x); acc[j].y = fmaf(d, x.y, acc[j].y); acc[j].z = fmaf(d, x.z, acc[j].z); acc[j].w = fmaf(d, x.w, acc[j].w);
    }
    if (NH == kNO) { s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w; }
    const f32x4 dz = f32x4{s0.x * lrelu_mask(x.x), s0.y * lrelu_mask(x.y), s0.z * lrelu_mask(x.z), s0.w * lrelu_mask(x.w)};
    if (a.dZ != nullptr) *reinterpret_cast<f32x4*>(a.dZ + (size_t)m * a.H + k0) = dz;
    if (b.dZ16 != nullptr) {
      const h4 hz = h4{(_Float16)(dz.x * b.scale16), (_Float16)(dz.y * b.scale16), (_Float16)(dz.z * b.scale16), (_Float16)(dz.w * b.scale16)};
      *reinterpret_cast<h4*>(b.dZ16 + (size_t)m * a.H + k0) = hz;
    }
  }
  if (want_w) {
#pragma unroll
    for (int j = 0; j < NH; ++j) *reinterpret_cast<f32x4*>(s_red + ((w * NH + j) * 256 + lane * 4)) = acc[j];
  }
  if (!want_w) return;
  __syncthreads();
  float* slab = b.slab2 + (size_t)blockIdx.x * NH * a.H;
  for (int i = tid; i < NH * 256; i += 256) {
    const int j = i >> 8, c = i & 255;
    slab[(size_t)j * a.H + kb + c] = (s_red[(0 * NH + j) * 256 + c] + s_red[(1 * NH + j) * 256 + c]) +
                                     (s_red[(2 * NH + j) * 256 + c] + s_red[(3 * NH + j) * 256 + c]);
  }
  if (blockIdx.y == 0 && tid < NH) {
    float v = 0.0f;
    for (int m = 0; m < 64; ++m) v += s_dy[m * NH + tid];
    const size_t n_chunks = b.chunks > 0 ? (size_t)b.chunks : (size_t)gridDim.x;      // (rider blocks extend the grid beyond the row chunks)
    b.slab2[n_chunks * NH * a.H + blockIdx.x * 16 + tid] = v;
  }
}
// adds the row-chunk slabs of k_head_bwd_big: block = (64 columns, head j); the 4 waves take every
// 4th chunk and are combined in fixed order.  Writes dW, db and one sum-of-squares partial per
// (head, 64 columns) into partial[j * H/64 + column block].
template <int NH>
__global__ __launch_bounds__(256) void k_head_wred(HeadBwdBigArgs b, int chunks) {
  const HeadBwdArgs& a = b.a;
  __shared__ float s[4][64];
  __shared__ float s_bias;
  const int j = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 64 + lane;
  float v = 0.0f;
#pragma unroll 4
  for (int c = w; c < chunks; c += 4) v += b.slab2[((size_t)c * NH + j) * a.H + k];
  s[w][lane] = v;
  if (w == 1) {
    float t = 0.0f;
    if (blockIdx.x == 0) {
      for (int c = lane; c < chunks; c += 64) t += b.slab2[(size_t)chunks * NH * a.H + c * 16 + j];
      t = wave_sum64(t);
      if (lane == 0) a.db[j] = t;
    }
    if (lane == 0) s_bias = t * t;
  }
  __syncthreads();
  if (w != 0) return;
  v = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
  a.dW[(size_t)j * a.H + k] = v;
  float ssq = v * v;
  ssq = wave_sum64(ssq);
  if (lane == 0 && a.partial != nullptr) a.partial[j * gridDim.x + blockIdx.x] = ssq + s_bias;
}

// ---- optimiser -----------------------------------------------------------------
// Sum of squares of a gradient arena -> per-block partials (used after an
// all-reduce, where the GEMM-epilogue partials no longer describe the reduced
// gradient).
static __global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, size_t n4,
                                               float* __restrict__ partial) {
  __shared__ float s[4];
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// Data-parallel exchange in half the bytes (dqnhip_dp_init flag DQNHIP_DP_HALF_GRADS): the gradient arena crosses
// the links as bf16 (fp32's exponent range: no loss scale, no overflow; 8 significant bits, round-to-nearest-even)
// and is widened again by the pass that takes the clip norm of the reduced gradient.
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static __global__ __launch_bounds__(256) void k_to_bf16(const float* __restrict__ g, size_t n4, uint16_t* __restrict__ out) {
  typedef __attribute__((ext_vector_type(4))) uint16_t u16x4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    reinterpret_cast<u16x4*>(out)[i] = u16x4{f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
  }
}
// bf16 image (after the all-reduce) -> fp32 arena + sum-of-squares partials (k_sumsq's layout and order)
static __global__ __launch_bounds__(256) void k_sumsq_bf16(const uint16_t* __restrict__ in, float* __restrict__ g, size_t n4,
                                                    float* __restrict__ partial) {
  typedef __attribute__((ext_vector_type(4))) uint16_t u16x4;
  __shared__ float s[4];
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const u16x4 h = reinterpret_cast<const u16x4*>(in)[i];
    const f32x4 v = f32x4{__builtin_bit_cast(float, (uint32_t)h.x << 16), __builtin_bit_cast(float, (uint32_t)h.y << 16),
                          __builtin_bit_cast(float, (uint32_t)h.z << 16), __builtin_bit_cast(float, (uint32_t)h.w << 16)};
    reinterpret_cast<f32x4*>(g)[i] = v;
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// SGDSolver::ClipGradients + AdamSolver::ComputeUpdateValue + Net::Update +
// DQN::SoftUpdateNet in ONE pass over (w, g, m, v, w_target)
// (Caffe sgd_solver.cpp/adam_solver.cpp @2ef5847; src/dqn.cpp:
// 904, 964, 967-970, 1085-1096).  36 B/param of HBM traffic instead of Caffe's
// ~7 separate param-sized passes plus the separate soft-update pass.
struct TickArgs {
  DevState* st; float* critic_tail; float* actor_tail;
  const float* loss_partial; int n_loss; const double* q_partial; int n_q; float batch;
  // host-mapped (pinned) copy of {critic_loss, avg_q, flags}: written by the update's last block, so
  // that dqnhip_read_stats needs a stream sync but no device-to-host copy (null: none)
  float* host_stats;
};
struct AdamArgs {
  float* w; float* g; float* m; float* v; float* wt;
  _Float16* w16; _Float16* wt16;             // fp16 mode: fp16 mirrors of w / wt, same offsets (null otherwise)
  float* w_sh; float* wt_sh; size_t n4_sh;   // float4 [0, n4_sh) of w / wt live in another learner's arena (ShareParameters)
  size_t n4;                      // arena length / 4
  size_t skip4;                   // the strided pass starts here: float4 [0, skip4) belong to the launch's first-layer riders (0: none)
  const float* partial; int n_partial;
  const float* corr_pre;          // this step's bias correction, evaluated earlier in the update (DevState::adam_corr); null: here
  const int* soft_pre;            // with corr_pre: this update's soft-update switch (DevState::soft_now)
  float lr, beta1, beta2, eps, clip, tau;
  int soft_update_freq;
  int which;                      // 0 actor, 1 critic (selects the iter counter)
  DevState* st;
  // the update's last launch also does k_tick's work: the block that finishes last (arrival
  // ticket; no fence needed — it consumes nothing the other blocks of THIS launch produced, and
  // by then every block has read the iteration counters it is about to advance) runs tick_body
  int tick_on;                    // 1: block 0 also runs tick_body (requires corr_pre / soft_pre)
  TickArgs tick;
};
// body shared by the stand-alone kernel and the mixed GEMM+Adam launch: block `blk` of
// `nblk` 256-thread blocks strides over the arena slice
// per-launch scalars of the optimiser pass into s[4..7]: clip scale, lr * Adam correction, soft-update
// switch, skip flag.  Every block re-derives them from the same partials in the same order.
// PRE: the caller guarantees corr_pre / soft_pre (inside an update) — the stand-alone path's two double pow() are not compiled in
template <bool PRE = false>
__device__ __forceinline__ void adam_scalars(const AdamArgs& a, int blk, float* s /*>= 8 floats*/) {
  // every block re-derives the same global L2 norm from the partials, in the
  // same order -> bit-identical scale everywhere, no extra launch
  float acc = 0.0f;
  for (int i = threadIdx.x; i < a.n_partial; i += 256) acc += a.partial[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  // one lane per block evaluates the per-launch scalars (two double pow() are ~500 instructions: run by
  // every thread they made this kernel VALU-bound: 733 VALU instructions per wave, 8 waves per SIMD)
  if (threadIdx.x == 0) {
    const float sumsq = (s[0] + s[1]) + (s[2] + s[3]);
    const float l2 = sqrtf(sumsq);
    s[4] = (a.clip >= 0.0f && l2 > a.clip) ? a.clip / l2 : 1.0f;
    if (PRE || a.corr_pre != nullptr) {                  // inside an update: both were left in DevState by its first launch
      s[5] = a.lr * *a.corr_pre;
      s[6] = *a.soft_pre ? 1.0f : 0.0f;
    } else {
      const int it_a = a.st->actor_iter, it_c = a.st->critic_iter;
      const int t = (a.which == 0 ? it_a : it_c) + 1;    // t = iter_ + 1 (before increment)
      s[5] = a.lr * adam_correction(a.beta1, a.beta2, t);
      // soft update condition uses max_iter() AFTER both increments (src/dqn.cpp:967)
      const int mx = (it_a + 1) > (it_c + 1) ? (it_a + 1) : (it_c + 1);
      s[6] = ((mx % a.soft_update_freq) == 0) ? 1.0f : 0.0f;
    }
    // A non-finite norm (fp16 mode: an overflowed dZ panel) would give scale = clip/inf = 0 and
    // g*0 = NaN in m, v, w and the targets for good.  Every block derives the same norm, so every
    // block takes the same decision: skip the whole step and raise the sticky flag.
    s[7] = isfinite(sumsq) ? 0.0f : 1.0f;
    if (s[7] != 0.0f && blk == 0) { atomicOr(&a.st->flags, kFlagGradNorm); atomicAdd(&a.st->skipped_steps, 1); }   // one launch per net per update
  }
  __syncthreads();
}
// one float4 of the optimiser step, in place (the strided pass and the first-layer riders share it: same expression, same bits)
__device__ __forceinline__ void adam_apply4(const AdamArgs& a, float scale, float step, bool soft, const f32x4& g, f32x4& m, f32x4& v, f32x4& w, f32x4& wt) {
  const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
  const float tau = a.tau, omt = 1 - a.tau;
  const float* gp = reinterpret_cast<const float*>(&g); float* mp = reinterpret_cast<float*>(&m);
  float* vp = reinterpret_cast<float*>(&v); float* wp = reinterpret_cast<float*>(&w);
  float* tp = reinterpret_cast<float*>(&wt);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gi = gp[e] * scale;
    const float mi = fmaf(omb1, gi, a.beta1 * mp[e]);
    const float vi = fmaf(omb2, gi * gi, a.beta2 * vp[e]);
    const float upd = step * (mi / (sqrtf(vi) + a.eps));
    const float wi = wp[e] - upd;
    mp[e] = mi; vp[e] = vi; wp[e] = wi;
    if (soft) tp[e] = fmaf(tau, wi, omt * tp[e]);
  }
}
// ... and one element (a first-layer rider's bias: one per thread)
__device__ __forceinline__ void adam_apply1(const AdamArgs& a, float scale, float step, bool soft, float g, float& m, float& v, float& w, float& wt) {
  const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
  const float tau = a.tau, omt = 1 - a.tau;
  const float gi = g * scale;
  const float mi = fmaf(omb1, gi, a.beta1 * m);
  const float vi = fmaf(omb2, gi * gi, a.beta2 * v);
  const float upd = step * (mi / (sqrtf(vi) + a.eps));
  const float wi = w - upd;
  m = mi; v = vi; w = wi;
  if (soft) wt = fmaf(tau, wi, omt * wt);
}
template <int U = 1, int NT = 0, bool PRE = false>
__device__ __forceinline__ void adam_soft_body(const AdamArgs& a, int blk, int nblk, float* s /*>= 8 floats*/) {
  // U float4 per array in flight per thread (U * 5 x 16-B loads before the first use); NT: the gradient is
  // read exactly once per update and never again -> non-temporal
  f32x4 g[U], m[U], v[U], w[U], wt[U];
  f32x4* wq[U]; f32x4* tq[U];
  auto load = [&](size_t i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * 256;
      if (i < a.n4) {
        wq[u] = reinterpret_cast<f32x4*>(i < a.n4_sh ? a.w_sh : a.w) + i;
        tq[u] = reinterpret_cast<f32x4*>(i < a.n4_sh ? a.wt_sh : a.wt) + i;
        g[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g) + i) : reinterpret_cast<const f32x4*>(a.g)[i];
        m[u] = reinterpret_cast<const f32x4*>(a.m)[i];
        v[u] = reinterpret_cast<const f32x4*>(a.v)[i];
        w[u] = *wq[u];
        wt[u] = *tq[u];
      }
    }
  };
  // The first (for most threads: the only, or one of two) batch of loads goes out BEFORE the per-launch scalars are
  // derived: none of them depends on the clip scale, and the scalars' own chain (partials from the L2 of other XCDs ->
  // wave sums -> barrier -> sqrt / divide in one lane -> barrier) is ~1.5 us that every block would otherwise spend
  // with nothing in flight.
  size_t i0 = a.skip4 + (size_t)blk * (256 * U) + threadIdx.x;
  if (i0 < a.n4) load(i0);
  adam_scalars<PRE>(a, blk, s);
  if (s[7] != 0.0f) return;
  const float scale = s[4];
  const float step = s[5];
  const bool soft = s[6] != 0.0f;
  for (bool first = true; i0 < a.n4; i0 += (size_t)nblk * (256 * U), first = false) {
    if (!first) load(i0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * 256;
      if (i >= a.n4) continue;
      adam_apply4(a, scale, step, soft, g[u], m[u], v[u], w[u], wt[u]);
      const float* wp = reinterpret_cast<const float*>(&w[u]); const float* tp = reinterpret_cast<const float*>(&wt[u]);
      reinterpret_cast<f32x4*>(a.m)[i] = m[u];
      reinterpret_cast<f32x4*>(a.v)[i] = v[u];
      *wq[u] = w[u];
      if (soft) *tq[u] = wt[u];
      if (a.w16 != nullptr) {
        typedef __attribute__((ext_vector_type(4))) _Float16 h16x4_t;
        reinterpret_cast<h16x4_t*>(a.w16)[i] = h16x4_t{(_Float16)wp[0], (_Float16)wp[1], (_Float16)wp[2], (_Float16)wp[3]};
        if (soft) reinterpret_cast<h16x4_t*>(a.wt16)[i] = h16x4_t{(_Float16)tp[0], (_Float16)tp[1], (_Float16)tp[2], (_Float16)tp[3]};
      }
    }
  }
}
template <int U, int NT>
__global__ __launch_bounds__(256) void k_adam_soft_t(AdamArgs a) {
  __shared__ float s[8];
  adam_soft_body<U, NT>(a, blockIdx.x, gridDim.x, s);
}
__device__ __forceinline__ void tick_body(const TickArgs& a, float* sdot, double* sq, bool skipped_now);   // below

static __global__ __launch_bounds__(256) void k_adam_soft(AdamArgs a) {
  __shared__ float s[8];
  __shared__ double sq[4];
  adam_soft_body<1, 0>(a, blockIdx.x, gridDim.x, s);
  // The update's bookkeeping (statistics, ++iter of both solvers, the sampling counter) rides in block 0 of the update's
  // last launch.  It used to wait for the LAST block to arrive (two levels of arrival counters: every block read the
  // iteration counters in its prologue) — a tail of 2.5-5 us behind the last stores (same-box A/B: 3 108-3 132 against
  // 3 138-3 157 updates/s without it).  With the bias correction and the soft-update switch left in DevState by the
  // update's first launch, no block of this launch reads anything tick_body writes, so block 0 runs it as soon as its
  // own slice is done, beside the other 2047 blocks.
  if (a.tick_on && blockIdx.x == 0) {
    const bool skipped = s[7] != 0.0f;
    __syncthreads();
    tick_body(a.tick, s, sq, skipped);
  }
}

// The update's last launch inside a multi-update graph: the first g.blocks workgroups are the NEXT update's gather
// (nothing it reads or writes is touched by this optimiser pass: the minibatch panels are dead until that update's first
// forward, its scalars go to the other DevState slot, its counters come from DevState::gbase), the rest is k_adam_soft.
// Takes the gather (a ~5-us launch of two dependent HBM round trips) off the chain of all but the first update of such a graph.
static __global__ __launch_bounds__(256) void k_adam_soft_gather(AdamArgs a, GatherArgs g) {
  __shared__ float s[8];
  __shared__ double sq[4];
  if ((int)blockIdx.x < g.blocks) { gather_block(g, (int)blockIdx.x); return; }
  const int blk = (int)blockIdx.x - g.blocks;
  adam_soft_body<1, 0>(a, blk, (int)gridDim.x - g.blocks, s);
  if (a.tick_on && blk == 0) {
    const bool skipped = s[7] != 0.0f;
    __syncthreads();
    tick_body(a.tick, s, sq, skipped);
  }
}

// ---- the first tower layer of the pass that follows, inside the optimiser launch (round 5) -------------------------------------
// The launch after the critic's optimiser pass is the first layer of critic(s, mu(s)): K = S + 10 padded to 64 / 128, a 5-us
// launch-floor link whose only late operand is the layer's own, just-updated weights.  Here the launch's first r.blocks
// workgroups each OWN 16 output rows of W1 (and their 16 biases): they take the step on that slice (adam_apply4: the strided pass's
// arithmetic), keep the new weights in LDS and run the layer for those 16 outputs over every minibatch row — the split of the
// reduction over the four waves, the step order and the (w0 + w1) + (w2 + w3) reduction of fwd_direct_body, so the activations
// are bit-identical to that launch's.  The strided pass belongs to the other workgroups and starts behind the slice
// (AdamArgs::skip4).  The riders are issued first and done after ~10 us of a 20-us pass.
// (First form, measured: the riders also took their share of the strided pass and walked the rows one 16-row tile at a time,
// every step behind its own load round trip: 26.8 us per launch against 19.7 + 5.0.)
struct FirstLayerRider {
  const float* X; int ldx;      // [rows][Kp]: the layer's input panel, complete before this launch
  float* Y; int ldy;            // [rows][N] out
  int rows, Kp, N;              // Kp = 64 G, rows % 16 == 0, N % 16 == 0; W1 = arena float4 [0, N Kp / 4), b1 behind it
  int blocks;                   // N / 16
};
// One step of a first-layer rider: the four 16-row tiles [t4, t4 + 4) of outputs [out0, out0 + 16) — fwd_direct_body's arithmetic,
// element for element (the reduction split over the four waves, its step order, (w0 + w1) + (w2 + w3)).  pw: this lane's weight
// fragments (row li of the 16, k = wave Kw + 16 kb + 4 lg; LDS or global); qf: the tiles' operands, requested earlier; the
// operands of step t4 + 4 are requested into qf behind the MFMAs; wave w reduces tile t4 + w.  bias16: the 16 outputs' biases
// (LDS or global; null: none).  One 16-KB parking area per workgroup (six workgroups per CU must keep fitting the LDS).
template <int G>
__device__ __forceinline__ void l0_step(const float* pw, const float* xq, int ldx, f32x4 (&qf)[4][G], int t4, int T, const float* bias16, bool relu,
                                        float* Y, int ldy, int out0, float* park, bool first) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  f32x4 acc[4], pf[G];
#pragma unroll
  for (int kb = 0; kb < G; ++kb) pf[kb] = *reinterpret_cast<const f32x4*>(pw + kb * 16);      // (re-read per step: 8 VGPRs the 80-register budget does not have)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < G; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[j] = DQN_MFMA(pf[kb][s], qf[j][kb][s], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kb = 0; kb < G; ++kb)
      qf[j][kb] = *reinterpret_cast<const f32x4*>(xq + (size_t)(t4 + 4 + j < T ? t4 + 4 + j : T - 1) * 16 * ldx + kb * 16);   // (beyond the last tile: a valid row, unused)
  f32x4* pk = reinterpret_cast<f32x4*>(park);
  if (!first) __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) pk[(j * 4 + wave) * 64 + lane] = acc[j];
  __syncthreads();
  if (t4 + wave < T) {
    const f32x4* pj = pk + wave * 256;
    const f32x4 a0 = pj[lane], a1 = pj[64 + lane], a2 = pj[128 + lane], a3 = pj[192 + lane];
    f32x4 o;
    o.x = (a0.x + a1.x) + (a2.x + a3.x); o.y = (a0.y + a1.y) + (a2.y + a3.y);
    o.z = (a0.z + a1.z) + (a2.z + a3.z); o.w = (a0.w + a1.w) + (a2.w + a3.w);
    if (bias16 != nullptr) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias16 + lg * 4);
      o.x += bias.x; o.y += bias.y; o.z += bias.z; o.w += bias.w;
    }
    if (relu) { o.x = lrelu_fwd(o.x); o.y = lrelu_fwd(o.y); o.z = lrelu_fwd(o.z); o.w = lrelu_fwd(o.w); }
    *reinterpret_cast<f32x4*>(Y + (size_t)((t4 + wave) * 16 + li) * ldy + out0 + (lg << 2)) = o;
  }
}
template <int G>
struct FirstLayerWork {
  const AdamArgs& a; const FirstLayerRider& r; const int blk; float* sW; float* sB; float* park;
  __device__ __forceinline__ FirstLayerWork(const AdamArgs& a_, const FirstLayerRider& r_, int blk_, float* sW_, float* sB_, float* park_)
      : a(a_), r(r_), blk(blk_), sW(sW_), sB(sB_), park(park_) {}
  static constexpr int NH = G == 1 ? 4 : 0;    // (G = 2: the step's own operands are 45 of the 80 registers)
  f32x4 g[G], m[G], v[G], w[G], wt[G], qf[4][G];
  float bg, bm, bv, bw, bwt;
  __device__ __forceinline__ size_t widx(int u) const { return ((size_t)blk * G + u) * 256 + threadIdx.x; }
  __device__ __forceinline__ size_t bidx() const { return (size_t)r.N * r.Kp + (size_t)blk * 16 + threadIdx.x; }
  // everything the step on this workgroup's slice reads, requested in ONE round trip before the launch's scalars are derived
  // (the memory system is saturated by the strided pass beside it: every dependent round trip costs ~3 us here)
  __device__ __forceinline__ void request() {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const size_t i = widx(u);
      g[u] = reinterpret_cast<const f32x4*>(a.g)[i]; m[u] = reinterpret_cast<const f32x4*>(a.m)[i];
      v[u] = reinterpret_cast<const f32x4*>(a.v)[i]; w[u] = reinterpret_cast<const f32x4*>(a.w)[i]; wt[u] = reinterpret_cast<const f32x4*>(a.wt)[i];
    }
    if (threadIdx.x < 16) { const size_t i = bidx(); bg = a.g[i]; bm = a.m[i]; bv = a.v[i]; bw = a.w[i]; bwt = a.wt[i]; }
    // ... and the first row tiles' operands of the layer (as many as the 80-register budget holds beside the step's operands)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int T = r.rows >> 4;
    const float* xq = r.X + (size_t)li * r.ldx + wave * (r.Kp >> 2) + lg * 4;
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
      for (int kb = 0; kb < G; ++kb) qf[j][kb] = *reinterpret_cast<const f32x4*>(xq + (size_t)(j < T ? j : T - 1) * 16 * r.ldx + kb * 16);
  }
  __device__ __forceinline__ void run(float scale, float step, bool soft, bool apply) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int Kw = r.Kp >> 2, T = r.rows >> 4;
    if (apply) {
#pragma unroll
      for (int u = 0; u < G; ++u) adam_apply4(a, scale, step, soft, g[u], m[u], v[u], w[u], wt[u]);
      if (threadIdx.x < 16) adam_apply1(a, scale, step, soft, bg, bm, bv, bw, bwt);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const size_t i = widx(u);
      if (apply) {
        reinterpret_cast<f32x4*>(a.m)[i] = m[u]; reinterpret_cast<f32x4*>(a.v)[i] = v[u]; reinterpret_cast<f32x4*>(a.w)[i] = w[u];
        if (soft) reinterpret_cast<f32x4*>(a.wt)[i] = wt[u];
      }
      reinterpret_cast<f32x4*>(sW)[u * 256 + threadIdx.x] = w[u];
    }
    if (threadIdx.x < 16) {
      const size_t i = bidx();
      if (apply) { a.m[i] = bm; a.v[i] = bv; a.w[i] = bw; if (soft) a.wt[i] = bwt; }
      sB[threadIdx.x] = bw;
    }
    const float* xq = r.X + (size_t)li * r.ldx + wave * Kw + lg * 4;
#pragma unroll
    for (int j = NH; j < 4; ++j)          // (the rest of the first group: behind the step's stores, whose registers they take over)
#pragma unroll
      for (int kb = 0; kb < G; ++kb) qf[j][kb] = *reinterpret_cast<const f32x4*>(xq + (size_t)(j < T ? j : T - 1) * 16 * r.ldx + kb * 16);
    __syncthreads();
    // the layer: outputs [16 blk, +16) x every row, four row tiles per step (l0_step)
    const float* pw = sW + li * r.Kp + wave * Kw + lg * 4;
    for (int t4 = 0; t4 < T; t4 += 4) l0_step<G>(pw, xq, r.ldx, qf, t4, T, sB, true, r.Y, r.ldy, blk * 16, park, t4 == 0);
  }
};
template <int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_adam_soft_fwd1(AdamArgs a, FirstLayerRider r) {   // (six workgroups per CU, as k_adam_soft: 1536 resident at once)
  __shared__ float s[8];
  __shared__ __attribute__((aligned(16))) float sW[16 * 64 * G];
  __shared__ __attribute__((aligned(16))) float sB[16];
  __shared__ __attribute__((aligned(16))) float park[4096];
  if ((int)blockIdx.x < r.blocks) {
    FirstLayerWork<G> work(a, r, (int)blockIdx.x, sW, sB, park);
    work.request();
    adam_scalars<true>(a, -1, s);          // (-1: the strided pass's first workgroup reports a skipped step)
    // (a skipped step — non-finite gradient norm — still runs the layer, on the weights as they are)
    work.run(s[4], s[5], s[6] != 0.0f, s[7] == 0.0f);
    return;
  }
  adam_soft_body<1, 0, true>(a, (int)blockIdx.x - r.blocks, (int)gridDim.x - r.blocks, s);
}

// ... with the NEXT update's gather in the same launch (multi-update graphs, round 5: the gather moves from the update's last launch
// to this one, so that the next update's first-layer inputs are complete BEFORE the actor's optimiser launch — k_adam_soft_l0)
template <int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_adam_soft_fwd1_gather(AdamArgs a, FirstLayerRider r, GatherArgs g) {
  __shared__ float s[8];
  __shared__ __attribute__((aligned(16))) float sW[16 * 64 * G];
  __shared__ __attribute__((aligned(16))) float sB[16];
  __shared__ __attribute__((aligned(16))) float park[4096];
  const int b = (int)blockIdx.x;
  if (b < r.blocks) {
    FirstLayerWork<G> work(a, r, b, sW, sB, park);
    work.request();
    adam_scalars<true>(a, -1, s);
    work.run(s[4], s[5], s[6] != 0.0f, s[7] == 0.0f);
    return;
  }
  if (b < r.blocks + g.blocks) { gather_block(g, b - r.blocks); return; }
  adam_soft_body<1, 0, true>(a, b - r.blocks - g.blocks, (int)gridDim.x - r.blocks - g.blocks, s);
}

// ---- the NEXT update's four first tower layers inside the actor's optimiser launch (round 5) -------------------------------------
// With the next update's minibatch panels gathered one launch group earlier (above; the two panels this update still reads exist
// twice), nothing but the actor's own step stands between this launch and the next update's first GEMM launch
// (first_layers_launch).  The workgroups that own 16 rows of the ACTOR's W1 take the step in registers — which also gives them
// the target actor's soft-updated rows — and run actor(s) and actor_target(s') for those outputs (ActorL0); critic(s, a)'s first
// layer and the state half of critic_target's read weights that have been final since the critic's step: plain riders (PlainL0).
// Every element as fwd_direct_body computes it.  The next update then starts at its second layer.
struct ActorL0 {
  const float* Xs; const float* Xn; int ldx;   // the next update's state / next-state panels [rows][64]
  float* Ys; float* Yn; int ldy;               // actor(s), actor_target(s') first-layer activations
  int rows, N;                                 // Kp = 64 (one float4 of W1 per rider thread)
  int blocks;                                  // N / 16
};
struct PlainL0 {
  const float* W; int ldw; const float* bias;  // [N][ldw]; bias null: none (and no ReLU: a partial pre-activation)
  const float* X; int ldx; float* Y; int ldy;
  int rows, Kred, N;                           // Kred = 64 or 128 (<= ldw)
  float* xcopy_dst; int xcopy_col, xcopy_n;    // GemmProblem::xcopy_dst (null: none)
  int blocks;                                  // N / 16
};
struct ActorL0Work {
  const AdamArgs& a; const ActorL0& r; const int blk; float* sW; float* sB; float* park;
  __device__ __forceinline__ ActorL0Work(const AdamArgs& a_, const ActorL0& r_, int blk_, float* sW_, float* sB_, float* park_)
      : a(a_), r(r_), blk(blk_), sW(sW_), sB(sB_), park(park_) {}
  f32x4 g, m, v, w, wt, qs[4][1], qn[4][1];
  float bg, bm, bv, bw, bwt;
  __device__ __forceinline__ size_t widx() const { return (size_t)blk * 256 + threadIdx.x; }
  __device__ __forceinline__ size_t bidx() const { return (size_t)r.N * 64 + (size_t)blk * 16 + threadIdx.x; }
  __device__ __forceinline__ void request() {
    const size_t i = widx();
    g = reinterpret_cast<const f32x4*>(a.g)[i]; m = reinterpret_cast<const f32x4*>(a.m)[i];
    v = reinterpret_cast<const f32x4*>(a.v)[i]; w = reinterpret_cast<const f32x4*>(a.w)[i]; wt = reinterpret_cast<const f32x4*>(a.wt)[i];
    if (threadIdx.x < 16) { const size_t j = bidx(); bg = a.g[j]; bm = a.m[j]; bv = a.v[j]; bw = a.w[j]; bwt = a.wt[j]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int T = r.rows >> 4;
    const size_t x0 = (size_t)li * r.ldx + wave * 16 + lg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t x = x0 + (size_t)(j < T ? j : T - 1) * 16 * r.ldx;
      qs[j][0] = *reinterpret_cast<const f32x4*>(r.Xs + x);
    }
  }
  __device__ __forceinline__ void run(float scale, float step, bool soft, bool apply) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int T = r.rows >> 4;
    if (apply) {
      adam_apply4(a, scale, step, soft, g, m, v, w, wt);
      if (threadIdx.x < 16) adam_apply1(a, scale, step, soft, bg, bm, bv, bw, bwt);
      const size_t i = widx();
      reinterpret_cast<f32x4*>(a.m)[i] = m; reinterpret_cast<f32x4*>(a.v)[i] = v; reinterpret_cast<f32x4*>(a.w)[i] = w;
      if (soft) reinterpret_cast<f32x4*>(a.wt)[i] = wt;
      if (threadIdx.x < 16) { const size_t j = bidx(); a.m[j] = bm; a.v[j] = bv; a.w[j] = bw; if (soft) a.wt[j] = bwt; }
    }
    reinterpret_cast<f32x4*>(sW)[threadIdx.x] = w; reinterpret_cast<f32x4*>(sW)[256 + threadIdx.x] = wt;
    if (threadIdx.x < 16) { sB[threadIdx.x] = bw; sB[16 + threadIdx.x] = bwt; }
    const float* xs = r.Xs + (size_t)li * r.ldx + wave * 16 + lg * 4;
    const float* xn = r.Xn + (size_t)li * r.ldx + wave * 16 + lg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) qn[j][0] = *reinterpret_cast<const f32x4*>(xn + (size_t)(j < T ? j : T - 1) * 16 * r.ldx);   // (behind the step's stores, whose registers they take over)
    __syncthreads();
    const float* pw = sW + li * 64 + wave * 16 + lg * 4;
    for (int t4 = 0; t4 < T; t4 += 4) {      // the two nets take turns: each one's next operands are in flight under the other's step
      l0_step<1>(pw, xs, r.ldx, qs, t4, T, sB, true, r.Ys, r.ldy, blk * 16, park, t4 == 0);
      l0_step<1>(pw + 16 * 64, xn, r.ldx, qn, t4, T, sB + 16, true, r.Yn, r.ldy, blk * 16, park, false);
    }
  }
};
template <int G>
__device__ __forceinline__ void plain_l0_run(const PlainL0& p, int blk, float* park) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const int Kw = 16 * G, T = p.rows >> 4;
  const float* xq = p.X + (size_t)li * p.ldx + wave * Kw + lg * 4;
  f32x4 qf[4][G];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kb = 0; kb < G; ++kb) qf[j][kb] = *reinterpret_cast<const f32x4*>(xq + (size_t)(j < T ? j : T - 1) * 16 * p.ldx + kb * 16);
  if (p.xcopy_dst != nullptr && threadIdx.x < 16) {
    const float* src = p.W + (size_t)(blk * 16 + threadIdx.x) * p.ldw + p.xcopy_col;
    for (int j = 0; j < p.xcopy_n; ++j) p.xcopy_dst[(size_t)j * p.N + blk * 16 + threadIdx.x] = src[j];
  }
  const float* pw = p.W + (size_t)(blk * 16 + li) * p.ldw + wave * Kw + lg * 4;
  const float* b16 = p.bias != nullptr ? p.bias + blk * 16 : nullptr;
  for (int t4 = 0; t4 < T; t4 += 4) l0_step<G>(pw, xq, p.ldx, qf, t4, T, b16, p.bias != nullptr, p.Y, p.ldy, blk * 16, park, t4 == 0);
}
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_adam_soft_l0(AdamArgs a, ActorL0 ra, PlainL0 p0, PlainL0 p1) {
  __shared__ float s[8];
  __shared__ double sq[4];
  __shared__ __attribute__((aligned(16))) float sW[2 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float sB[32];
  __shared__ __attribute__((aligned(16))) float park[4096];
  int b = (int)blockIdx.x;
  if (b < ra.blocks) {
    ActorL0Work work(a, ra, b, sW, sB, park);
    work.request();
    adam_scalars<true>(a, -1, s);
    work.run(s[4], s[5], s[6] != 0.0f, s[7] == 0.0f);
    return;
  }
  b -= ra.blocks;
  if (b < p0.blocks) { if (p0.Kred == 64) plain_l0_run<1>(p0, b, park); else plain_l0_run<2>(p0, b, park); return; }
  b -= p0.blocks;
  if (b < p1.blocks) { if (p1.Kred == 64) plain_l0_run<1>(p1, b, park); else plain_l0_run<2>(p1, b, park); return; }
  b -= p1.blocks;
  adam_soft_body<1, 0, true>(a, b, (int)gridDim.x - ra.blocks - p0.blocks - p1.blocks, s);
  if (a.tick_on && b == 0) {
    const bool skipped = s[7] != 0.0f;
    __syncthreads();
    tick_body(a.tick, s, sq, skipped);
  }
}

// Sum of up to 8 co-located gradient arenas in rank order, written back to all (dqnhip_reduce_gradients_local)
struct LocalReduce { float* g[8]; int n; size_t n4; };
static __global__ __launch_bounds__(256) void k_local_reduce(LocalReduce a) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * 256) {
    f32x4 s = reinterpret_cast<const f32x4*>(a.g[0])[i];
    for (int r = 1; r < a.n; ++r) { const f32x4 v = reinterpret_cast<const f32x4*>(a.g[r])[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    for (int r = 0; r < a.n; ++r) reinterpret_cast<f32x4*>(a.g[r])[i] = s;
  }
}

// Sharded optimiser (DQNHIP_DP_SHARD_OPT): the sum of squares of THIS rank's slice of the reduced gradient, folded from
// k_sumsq's partials with the tree adam_scalars uses (strided sums, butterfly, fixed cross-wave order: a one-rank group
// then derives the same bits as the replicated form) into tail[3]; the 4-float tail is what the ranks all-reduce.
static __global__ __launch_bounds__(256) void k_shard_scal(const float* __restrict__ partial, int n_partial, float* tail) {
  __shared__ float s[4];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) tail[3] = (s[0] + s[1]) + (s[2] + s[3]);
}
// dqnhip_apply_update_sharded (one process standing in for every rank of a group in turn): total += the slice's sum
static __global__ void k_shard_accumulate(float* total, const float* tail, int first) {
  total[0] = first ? tail[3] : total[0] + tail[3];
}

// Reduce the per-block loss / q partials into the gradient-arena tails (tails_block, gemm_common.hip.h) in a launch of its own: the
// form for schedules without a carrier launch; otherwise the block rides in the net's last backward launch (TailsArgs::on).
static __global__ __launch_bounds__(256) void k_tails(TailsArgs a) {
  __shared__ float sdot[4];
  __shared__ double sq[4];
  tails_block(a, sdot, sq);
}

// End of update: publish (critic_loss, avg_q), advance both solver iterations
// (Step's ++iter_, set_iter(iter+1): src/dqn.cpp:904, 965) and the sampling counter.
// avg_q = std::accumulate(q, 0.0) / float(B) (src/dqn.cpp:915-916): the double sum
// is taken from the per-block double partials when they are local (single GPU),
// from the all-reduced float tail under data parallelism.
// One block of 256 threads: strided partial sums, fixed butterfly + fixed cross-wave order.
__device__ __forceinline__ void tick_body(const TickArgs& a, float* sdot /*[4]*/, double* sq /*[4]*/, bool skipped_now) {
  const int t = threadIdx.x;
  double qs = 0.0;
  if (a.q_partial != nullptr) {          // single GPU: reduce the per-block partials here
    float dot = 0.0f;
    for (int i = t; i < a.n_loss; i += 256) dot += a.loss_partial[i];
    for (int i = t; i < a.n_q; i += 256) qs += a.q_partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off, 64); qs += __shfl_xor(qs, off, 64); }
    if ((t & 63) == 0) { sdot[t >> 6] = dot; sq[t >> 6] = qs; }
    __syncthreads();
    if (t == 0) {
      dot = (sdot[0] + sdot[1]) + (sdot[2] + sdot[3]);
      qs = (sq[0] + sq[1]) + (sq[2] + sq[3]);
      a.critic_tail[0] = dot / a.batch / 2.0f; a.actor_tail[1] = (float)qs;     // EuclideanLoss: dot / num / 2
    }
  } else qs = (double)a.actor_tail[1];   // data parallel: tails were all-reduced
  if (t != 0) return;
  if (a.q_partial == nullptr && a.critic_tail[2] != 0.0f) atomicOr(&a.st->flags, kFlagTarget);   // some rank's target was not finite
  a.st->critic_loss = a.critic_tail[0];
  a.st->avg_q = (float)(qs / (double)a.batch);
  a.st->actor_iter += 1; a.st->critic_iter += 1; a.st->update_counter += 1;
  if (a.host_stats != nullptr) {
    // the flags were raised with device-scope atomics (by earlier kernels of this update, or by block 0 of THIS launch —
    // whose atomic may still be in flight: this block derived the same skip decision itself); read them the same way
    int fl = __hip_atomic_load(&a.st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (skipped_now) fl |= kFlagGradNorm;
    a.host_stats[0] = a.critic_tail[0]; a.host_stats[1] = (float)(qs / (double)a.batch);
    a.host_stats[2] = __builtin_bit_cast(float, fl);
  }
}
// ++iter of one solver (dqnhip_apply_update: set_iter(iter() + 1), src/dqn.cpp:965)
// It is also this path's "tick": an optimiser pass outside an update may have raised kFlagGradNorm (skipped step), and
// dqnhip_read_stats only reads the host-mapped words — mirror the sticky flags there (loss / avg_q stay the last update's).
static __global__ void k_advance_iter(DevState* st, int which, float* host_stats) {
  if (which == 0) st->actor_iter += 1; else st->critic_iter += 1;
  if (host_stats != nullptr) {
    const int fl = __hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    host_stats[2] = __builtin_bit_cast(float, fl);
  }
}

// ---- acting-time helpers ---------------------------------------------------------
// dense [n][S] -> padded panel [npad][SP] (pad rows/cols zero)
static __global__ void k_pack_rows(const float* __restrict__ src, int n, int S, float* __restrict__ dst,
                            int npad, int SP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad * SP) return;
  const int r = i / SP, c = i % SP;
  dst[i] = (r < n && c < S) ? src[(size_t)r * S + c] : 0.0f;
}
// critic input panel from dense states + dense actor outputs
static __global__ void k_pack_critic(const float* __restrict__ s, const float* __restrict__ a, int n, int S,
                              float* __restrict__ dst, int npad, int KP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad * KP) return;
  const int r = i / KP, c = i % KP;
  float v = 0.0f;
  if (r < n) { if (c < S) v = s[(size_t)r * S + c]; else if (c < S + kNO) v = a[(size_t)r * kNO + (c - S)]; }
  dst[i] = v;
}
static __global__ void k_unpack_out(const float* __restrict__ out16, int n, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * kNO) return;
  dst[i] = out16[(size_t)(i / kNO) * kAP + (i % kNO)];
}

}  // namespace dqnhip
