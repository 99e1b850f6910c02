// hgemm.hip.h — fp16-input / fp32-accumulate MFMA GEMM family for the mixed-precision learner
// (BASELINE.json config #5: "minibatch 4096, fp16 MFMA with fp32 accumulate").
//
// One kernel shape serves the three uses of a tower layer (reference: Caffe InnerProduct
// forward / backward, called from src/dqn.cpp:904, 923, 963):
//
//     C[m][n] = sum_k A[m][k] * B[n][k]
//
// Every operand lives in HBM in ONE orientation — the batch-major fp16 panels X16 [rows][k_in], dY16 [rows][n_out] and
// the fp16 weight mirror W16 [n_out][k_in] — and is either k-major for a GEMM (the reduction index contiguous: a lane's
// v_mfma_f32_32x32x16_f16 fragment = 8 consecutive k of one row = one 16-B ds_read_b128) or REDUCTION-major (the
// rows ARE the reduction index: the fragment then comes out of LDS through two transposing reads ds_read_b64_tr_b16):
//
//   FWD    m = batch row, n = out unit, k = in unit    A = X16 (k-major)            B = W16 (k-major)
//   DGRAD  m = batch row, n = in unit,  k = out unit   A = dY16 (k-major)           B = W16 (reduction-major)
//   WGRAD  m = out unit,  n = in unit,  k = batch row  A = dY16 (reduction-major)   B = X16 (reduction-major)
//
// (HGemm::ta / tb; rounds 1-2a kept every panel in both orientations instead.)
//
// Data path: global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no
// VGPR staging), 3-4 stage ring, one s_barrier per 64-deep K tile, counted vmcnt so the next
// tiles stay in flight across the barrier.  The LDS image is [rows][64 halves] = 128-B rows
// whose 16-B chunks are XOR-swizzled with (row>>1)&7 (reduction-major images: ((row>>1)&1)<<2): the DMA writes
// lane-linearly, so the swizzle is applied to the per-lane SOURCE address; the ds_read_b128 lane groups of a 32-row
// fragment ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-B slots of the 256-B bank row.
// Wave tile 64x64 (2x2 MFMA 32x32x16): 4 ds_read_b128 per 4 MFMA = 50 % of the LDS read rate.
//   <2,2>: 128x128 workgroup tile, waves 2x2.
//   <4,2>: 256x128 on eight waves (two per SIMD): two k-major problems of 4096 rows in one launch.
//   <1,1>:  64x64  workgroup tile, the 4 waves split each K tile (k16 slice w) and the four
//           partial tiles are added in the fixed order (w0+w1)+(w2+w3) — small minibatches.
// One launch carries up to four problems of one tile configuration (hgemm_nt), or all wgrads of a net plus the
// bias-gradient column sums (hgemm_group_db).
// Epilogue through LDS (fp32 tile): bias + leaky-ReLU / ReLU' mask / scale, then coalesced
// stores of the m-major fp16 panel and / or the fp32 panel (+ a per-workgroup sum of squares for the clip norm).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "gemm_common.hip.h"      // wave_sum64, TailsArgs (a rider block of hgemm_group_db)

namespace dqnhip {

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float hg_f32x16;
typedef __attribute__((ext_vector_type(4))) float hg_f32x4;

struct HGemm {
  const h16* A; int lda;
  const h16* B; int ldb;
  int M, N, K;                 // M % BM == 0, N % BN == 0, K % 64 == 0
  h16* C16; int ldc16;         // [M][ldc16]  m-major fp16 (may be null)
  h16* CT16; int ldct16;       // [N][ldct16] transposed fp16 (may be null)
  float* C32; int ldc32;       // [M][ldc32]  fp32 (may be null); columns >= n_valid32 are not written
  int n_valid32;
  const float* bias;           // [N] added before the activation (may be null)
  int relu;                    // leaky ReLU(0.01) on the result
  const h16* mask; int ldm;    // [M][ldm]: multiply by lrelu'(mask) = mask > 0 ? 1 : 0.01 (may be null)
  float scale32;               // the fp32 output is multiplied by this (loss-scale removal)
  float* sumsq_partial;        // one slot per workgroup of this problem: sum of squares of the fp32 values it wrote (may be null)
  // forward, top tower layer of the critic(s, mu(s)) pass (may be null): also write the seed of BackwardFrom(q_values_layer)
  // (src/dqn.cpp:918-923: q diff = -1 per row) taken through the head and this layer's ReLU, as the scaled fp16 panel the dgrad
  // chain reads: CS16[m][n] = fp16(((-seed_w[n]) * lrelu'(fp16(C[m][n]))) * seed_scale) — k_head_bwd(_big)<1>'s arithmetic
  // on the fp16-rounded activation, without its launch
  const float* seed_w; h16* CS16; int ldcs16; float seed_scale;
  // Operand orientation in memory.  0: k-major — [rows][ld] with the reduction index contiguous (a lane's MFMA fragment
  // is one 16-B piece).  1: REDUCTION-major — [K][ld] with the M (resp. N) index contiguous; fragments then come out of
  // LDS through the transposing read ds_read_b64_tr_b16.  With it the three layer GEMMs read the SAME batch-major panels
  // and the SAME weight mirror:  FWD  A = X[b][k] (0), B = W[n][k] (0);  DGRAD  A = dY[b][n] (0), B = W[n][k_in] (1: rows
  // are the reduction index n);  WGRAD  A = dY[b][n] (1), B = X[b][k] (1) — no transposed copy of anything.
  int ta, tb;
};

// Up to four independent problems of the same tile configuration in one launch (the two actors' /
// the two critics' same-depth layers; a layer's dgrad + wgrad at small minibatches; ALL wgrads of a net
// once its dgrad chain has produced every dZ panel): blocks [tile_end[i-1], tile_end[i]) work on g[i].
constexpr int kHGemmMax = 4;
struct HGemmBatch { HGemm g[kHGemmMax]; int n; int tile_end[kHGemmMax]; };

// bias gradients of all tower layers of one net: db_l[n] = scale * sum_b dY_l[b][n]
struct Db16 { const h16* dyt; int ld; int n_out; int rows; float* db; int row_base; };
struct Db16Batch { Db16 d[8]; int n; float scale; float* sumsq_partial; /* one slot per 64-column block of k_db16_cols (may be null) */ };

__device__ __forceinline__ int hg_tile_of_block(int bid, int total) {
  // XCD x (= bid % 8) gets a contiguous run of row-major tiles: they share A row panels and
  // sweep all of B, so each XCD's L2 holds its A panels + B once (bijective for any total)
  const int xcd = bid & 7, j = bid >> 3;
  const int q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// 2-D form for problems whose operand panels are LONG (wgrad: a panel is 128 columns x the whole minibatch, 1 MiB at
// 4096 rows): an XCD's 8 concurrent tiles form a 2 x 4 block of the tile grid, so its L2 pulls 2 A panels + 4 B panels
// instead of 1 + 8 (PMC, round 3: the grouped wgrad launch fetched 263 MB for ~50 MB of unique operands — 5.8 TB/s of
// fabric traffic over its 45 us, i.e. it ran AT the fabric rate).  Needs (tiles_m / 2) * (tiles_n / 4) % 8 == 0.
__device__ __forceinline__ bool hg_tile_2d(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  if ((tiles_m & 1) || (tiles_n & 3) || (((tiles_m >> 1) * (tiles_n >> 2)) & 7)) return false;
  const int xcd = bid & 7, j = bid >> 3;
  const int q = ((j >> 3) << 3) + xcd, jj = j & 7, bn = tiles_n >> 2;
  tm = ((q / bn) << 1) + (jj >> 2);
  tn = ((q % bn) << 2) + (jj & 3);
  return true;
}

#ifdef HG_CLOCKPROBE
__device__ unsigned long long hg_clk[2];   // shader cycles / 100 MHz ticks of block 17's main loop (test build only)
#endif

// TM: 32-row MFMA block PAIRS a wave stacks in M — wave tile (64 TM) x 64.  TM = 2 (<2,2,2>: four waves of 128 x 64 on a
// 256 x 128 workgroup tile): 6 fragment reads per 8 MFMAs instead of 4 per 4, and 48 KiB of operands per 4.2 MFLOP.
template <int WM, int WN, int TM = 1>
struct HGCfg {
  static constexpr int NW = (WM * WN >= 4) ? WM * WN : 4;                   // waves per workgroup (4, or 8 for <4,2>)
  static constexpr int NT = NW * 64;
  static constexpr int WK = (WM * WN >= 4) ? 1 : 4 / (WM * WN);
  static constexpr int BM = WM * 64 * TM, BN = WN * 64;
  // <2,2>: a stage is 256 row slots of 128 B (32 KiB): slots 0-127 = A rows, 128-255 = B rows of one 64-deep K tile.
  // <4,2>: 384 row slots (48 KiB): 256 A rows + 128 B rows.  <1,1>: two 64-deep sub-tiles, each 64 A rows + 64 B rows
  // (K step 128).
  static constexpr int KSTEP = (WK == 1) ? 64 : 128;
#ifndef HG_STAGES
#define HG_STAGES 4
#endif
  static constexpr int STAGE = (WK == 1) ? (BM + BN) * 128 : 32768;
  static constexpr int STAGES = (STAGE > 32768) ? 3 : HG_STAGES;
  static constexpr int LOADS = STAGE / 1024 / NW;                           // 1-KiB pieces per wave per stage (8; 6 for <4,2>)
  static constexpr int NSUB = (WK == 1) ? 4 : 2;                            // k16 steps per wave per stage
  static constexpr int TLD = BN + 4;                                        // fp32 epilogue tile row stride
  static constexpr int T_BYTES = WK * BM * TLD * 4;
  static constexpr int PIPE_BYTES = STAGES * STAGE;
  static constexpr int LDS_BYTES = PIPE_BYTES > T_BYTES ? PIPE_BYTES : T_BYTES;
};

// counted wait: at most N of this wave's vector-memory operations (the LDS-DMA pieces) outstanding
template <int N> __device__ __forceinline__ void hg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define HG_PIN() __builtin_amdgcn_sched_barrier(0)

typedef __attribute__((__vector_size__(4 * sizeof(short)))) short hg_s16x4;
typedef __attribute__((address_space(3))) hg_s16x4* hg_lds_s16x4_ptr;

// MODE = ta | tb << 1 of the problem (compile time: the fragment loads of the main loop differ)
template <int WM, int WN, int MODE, int TM = 1>
__device__ __forceinline__ void hgemm_body(const HGemm& g, const int bid) {
  using Cfg = HGCfg<WM, WN, TM>;
  constexpr int NW = Cfg::NW, NT = Cfg::NT;
  constexpr int MI = 2 * TM;                 // 32-row MFMA blocks per wave in M
  constexpr bool TA = (MODE & 1) != 0, TB = (MODE & 2) != 0;
  static_assert(MODE == 0 || (NW == 4 && TM == 1), "reduction-major operands: 128x128 and 64x64 tiles only");
  static_assert(TM == 1 || Cfg::WK == 1, "tall wave tiles: no in-workgroup split-K");
#ifdef HG_CLOCKPROBE
  const unsigned long long hg_c0 = clock64(), hg_r0 = wall_clock64();
#endif
  constexpr int WK = Cfg::WK, BM = Cfg::BM, BN = Cfg::BN, STAGES = Cfg::STAGES, TLD = Cfg::TLD;
  constexpr int NSUB = Cfg::NSUB, LOADS = Cfg::LOADS, KSTEP = Cfg::KSTEP;
  extern __shared__ __attribute__((aligned(1024))) unsigned char hg_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (WK == 1) ? (w >> 1) : 0, wn = (WK == 1) ? (w & 1) : 0;
  const int tiles_n = g.N / BN, tiles_m = g.M / BM;
  int tm, tn;
  if (!(TA && TB && hg_tile_2d(bid, tiles_m, tiles_n, tm, tn))) {
    const int T = hg_tile_of_block(bid, tiles_m * tiles_n);
    tm = T / tiles_n; tn = T % tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = g.K / KSTEP;

  // ---- LDS-DMA source addresses.  A piece is 8 row slots (1 KiB); lane -> (slot R, physical chunk pc) fetches logical
  // chunk pc ^ ((R>>1)&7).  A piece belongs to ONE matrix, so its source is a wave-uniform 64-bit base (SGPR pair) +
  // a 32-bit per-lane byte offset.
  //   4 waves: wave w owns pieces w*8 .. w*8+7 of every stage (waves {0,1}/{2,3} = A/B for <2,2>; even/odd for <1,1>).
  //   8 waves (<4,2>): wave w owns A pieces w*4 .. w*4+3 (A rows w*32 ..) and B pieces w*2, w*2+1 (B rows w*16 ..).
  const int r8 = lane >> 3, pc = lane & 7;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)hg_smem;
  const h16* wbase = nullptr; const h16* wbase_b = nullptr;
  size_t dma_step = KSTEP;                   // elements between two stages of this wave's source (KSTEP columns, or KSTEP rows)
  uint32_t voff[LOADS];
  constexpr int APW = BM / 8 / NW, BPW = BN / 8 / NW;      // per-wave A / B pieces of a stage in the rows-split-over-all-waves scheme
  if constexpr (NW == 4 && TM == 1) {
    const bool from_a = (WK == 1) ? (wu < 2) : ((wu & 1) == 0);
    const bool red = from_a ? TA : TB;                          // wave-uniform (a constant when TA == TB)
    const int ld = __builtin_amdgcn_readfirstlane(from_a ? g.lda : g.ldb);
    if (red) {
      // reduction-major operand: the wave's 64 slots are 64 consecutive REDUCTION rows of one 64-column strip —
      // <2,2>: strips {A cols 0-63, A cols 64-127, B cols 0-63, B cols 64-127} for waves 0..3; <1,1>: {A, B} x
      // {first, second 64-deep sub-tile}.  Same 128-B rows as the k-major image, but the swizzle that keeps the
      // transposing read conflict-free is chunk ^= ((row >> 1) & 1) << 2 (the four rows of a 32-lane cycle then
      // cover four different 64-B quarters of the bank row; PMC: 0 conflicts).
      wbase = (from_a ? g.A + m0 : g.B + n0) + ((WK == 1) ? (wu & 1) * 64 : 0) + (size_t)((WK == 1) ? 0 : (wu >> 1) * 64) * ld;
      dma_step = (size_t)(KSTEP * ld);
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        const int c = pc ^ (((r8 >> 1) & 1) << 2);
        voff[i] = (uint32_t)(((i * 8 + r8) * ld + c * 8) * 2);
      }
    } else {
      const int row_w = (WK == 1) ? (wu & 1) * 64 : 0;          // first matrix row (inside the tile) of this wave's slots
      wbase = (from_a ? g.A + (size_t)m0 * g.lda : g.B + (size_t)n0 * g.ldb) + (size_t)row_w * ld +
              ((WK == 1) ? 0 : (wu >> 1) * 64);
      dma_step = KSTEP;
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        const int R = (wu * LOADS + i) * 8 + r8;                // stage slot
        const int c = pc ^ ((R >> 1) & 7);
        voff[i] = (uint32_t)(((i * 8 + r8) * ld + c * 8) * 2);
      }
    }
  } else {
    // every wave owns a run of A rows and a run of B rows: <4,2> (8 waves) 32 + 16 rows, <2,2,2> (4 waves) 64 + 32
    static_assert(APW + BPW == LOADS, "stage pieces");
    wbase = g.A + (size_t)(m0 + wu * APW * 8) * g.lda;
    wbase_b = g.B + (size_t)(n0 + wu * BPW * 8) * g.ldb;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const bool pa = i < APW;
      const int rl = pa ? i * 8 : (i - APW) * 8;                // row inside this wave's run of the matrix
      const int R = (pa ? wu * APW * 8 : BM + wu * BPW * 8) + rl + r8;    // stage slot
      const int c = pc ^ ((R >> 1) & 7);
      voff[i] = (uint32_t)(((rl + r8) * (pa ? g.lda : g.ldb) + c * 8) * 2);
    }
  }
  // one piece of stage kt.  Inline asm: the compiler's waitcnt pass models a global_load_lds as a FLAT
  // operation that may touch LDS and then turns every later lgkmcnt wait into lgkmcnt(0), which
  // serialises the fragment prefetch; all ordering of the DMA is done by hand (counted vmcnt + barrier).
  auto issue1 = [&](int kt, int i) {
    const h16* base; uint32_t dst;
    if constexpr (NW == 4 && TM == 1) {
      base = wbase + (size_t)kt * dma_step;
      dst = lds0 + (uint32_t)((kt % STAGES) * Cfg::STAGE + (wu * LOADS + i) * 1024);
    } else {
      base = (i < APW ? wbase : wbase_b) + (size_t)kt * KSTEP;
      dst = lds0 + (uint32_t)((kt % STAGES) * Cfg::STAGE + (i < APW ? (wu * APW + i) : (BM / 8 + wu * BPW + (i - APW))) * 1024);
    }
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff[i]), "s"(base), "s"(dst) : "memory");
  };

  // ---- fragment read offsets (bytes inside a stage) for sub-step s
  const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
  int offa[NSUB], offb[NSUB];
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
    if constexpr (WK == 1) {
      const int o = (((2 * s + hi) ^ sw) << 4) + l31 * 128;
      offa[s] = wm * 64 * TM * 128 + o; offb[s] = (BM + wn * 64) * 128 + o;
    } else {
      const int o = (((2 * w + hi) ^ sw) << 4) + l31 * 128;     // this wave's k16 slice of each sub-tile
      offa[s] = s * 16384 + o; offb[s] = s * 16384 + 8192 + o;
    }
  }

  hg_f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // reduction-major image: a lane's fragment (8 reduction values of one output row/column) is two transposing
  // reads.  ds_read_b64_tr_b16 (semantics probed, profiles/r02_ds_read_tr_b16_probe.txt): within a 16-lane group
  // out[lane i][j] = in[lane 4j + (i >> 2)][i & 3], in[t] = the 4 halves at lane t's address.  So lane t = 4j + q
  // of group gq points at row (kb + j), columns 16 gq' + 4q of its 32-column block: the group then holds, per lane
  // i, column 16 gq' + i at rows kb .. kb+3.  Lanes 0-31 take reduction rows kb = k16 + 0 (+4 for the second read),
  // lanes 32-63 k16 + 8 (+12): the 32x32x16 operand layout.
  int tro[2][2] = {{0, 0}, {0, 0}};          // [read r][operand 0 = A, 1 = B]: byte offset inside a stage, block 0, sub-step 0
  if constexpr (TA || TB) {
    const int t16 = lane & 15, j4 = t16 >> 2, q4 = t16 & 3, gq = (lane >> 4) & 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = ((WK == 1) ? 0 : 16 * w) + hi * 8 + 4 * r + j4;      // + 16 s for <2,2> sub-steps (keeps (row >> 1) & 1)
      const int chunk = gq * 2 + (q4 >> 1);                                 // 16-B chunk inside the 64-B block
      const int o = row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4) + (q4 & 1) * 8;
      tro[r][0] = ((WK == 1) ? wm * 8192 : 0) + o;
      tro[r][1] = ((WK == 1) ? (BM + wn * 64) * 128 : 8192) + o;
    }
  }
  h16x8 fa[2][MI], fb[2][2];                // [register buffer][32-row block]
  auto load_frags = [&](int buf, int kt, int s) {
    const unsigned char* st = hg_smem + (kt % STAGES) * Cfg::STAGE;
    // reduction-major operand: block i (32 columns) = +64 B along the row, i.e. chunk bit 2: XOR 64 on the swizzled offset
    const uint32_t sbase = lds0 + (uint32_t)((kt % STAGES) * Cfg::STAGE + ((WK == 1) ? s * 16 * 128 : s * 16384));
    auto tr_frag = [&](int op, int i) -> h16x8 {
      union { hg_s16x4 h[2]; h16x8 v; } u;
#pragma unroll
      for (int r = 0; r < 2; ++r) u.h[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((hg_lds_s16x4_ptr)(sbase + (uint32_t)(tro[r][op] ^ (i * 64))));
      return u.v;
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (TA) fa[buf][i] = tr_frag(0, i);
      else fa[buf][i] = *reinterpret_cast<const h16x8*>(st + offa[s] + i * 32 * 128);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (TB) fb[buf][i] = tr_frag(1, i);
      else fb[buf][i] = *reinterpret_cast<const h16x8*>(st + offb[s] + i * 32 * 128);
    }
  };

  // ---- main loop.  Stages kt+1..kt+3 are in flight / landed while stage kt is multiplied; the barrier
  // at the top of iteration kt publishes stage kt+1 (so its first fragments can be fetched at the end
  // of iteration kt) and retires every wave's reads of stage kt-1, whose buffer stage kt+3 then reuses.
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) issue1(s, i);
    }
  {                                        // stage 0 landed: the other prologue stages may stay in flight
    const int newer = (nk < STAGES - 1 ? nk : STAGES - 1) - 1;
    if (newer >= 3) hg_wait_vm<3 * LOADS>(); else if (newer == 2) hg_wait_vm<2 * LOADS>(); else if (newer == 1) hg_wait_vm<LOADS>(); else hg_wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  load_frags(0, 0, 0);
  auto stage_body = [&](int kt, auto more_c) {
    constexpr bool more = decltype(more_c)::value;      // stage kt+3 exists: issue its pieces between the MFMAs
    if (more || kt + 1 < nk) {
      // stage kt+1 landed: stages kt+2 .. kt+STAGES-2 may stay in flight
      if constexpr (more) hg_wait_vm<(STAGES - 3) * LOADS>();
      else {
        const int newer = (nk - 2 - kt) < (STAGES - 3) ? (nk - 2 - kt) : (STAGES - 3);
        if (newer >= 2) hg_wait_vm<2 * LOADS>(); else if (newer == 1) hg_wait_vm<LOADS>(); else hg_wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int cur = s & 1;                // NSUB is even: register buffer 0 again at s = 0 of the next stage
      const int p_lo = s * LOADS / NSUB, p_hi = (s + 1) * LOADS / NSUB, p_mid = (p_lo + p_hi + 1) / 2;   // this sub-step's pieces
      HG_PIN();
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
      HG_PIN();
      if (s + 1 < NSUB) load_frags(cur ^ 1, kt, s + 1);
      else if (more || kt + 1 < nk) load_frags(cur ^ 1, kt + 1, 0);
      HG_PIN();
      if constexpr (more) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) if (i >= p_lo && i < p_mid) issue1(kt + STAGES - 1, i);
      }
      HG_PIN();
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
      HG_PIN();
      if constexpr (more) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) if (i >= p_mid && i < p_hi) issue1(kt + STAGES - 1, i);
      }
      HG_PIN();
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
#pragma unroll
      for (int i = 2; i < MI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][0], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][1], acc[i][1], 0, 0, 0);
      }
    }
  };
  int kt = 0;
  for (; kt + STAGES - 1 < nk; ++kt) stage_body(kt, std::true_type{});
  for (; kt < nk; ++kt) stage_body(kt, std::false_type{});
  __syncthreads();                          // pipeline buffers are dead: reuse them as the fp32 tile
#ifdef HG_CLOCKPROBE
  if (blockIdx.x == 17 && threadIdx.x == 0) { hg_clk[0] = clock64() - hg_c0; hg_clk[1] = wall_clock64() - hg_r0; }
#endif

  // dgrad: the ReLU' mask (the stored fp16 activation of the layer below, one 16-B piece per 8 outputs of this thread) is
  // fetched NOW, so that its latency runs under the accumulator -> LDS pass instead of inside the output loop, where each
  // of the thread's EIT iterations paid one exposed round trip (the dgrad launches cost 3.9 us more than the forward ones)
  constexpr int CPR = BN / 8;               // 8-output chunks per tile row
  constexpr int EIT = BM * CPR / NT;        // output-loop iterations of a thread
  static_assert(BM * CPR % NT == 0, "every thread runs the same number of output iterations");
  constexpr bool kMaskAhead = EIT <= 8;
  h16x8 mk_ahead[kMaskAhead ? EIT : 1];
  if constexpr (kMaskAhead) {
    if (g.mask) {
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int q = tid + it * NT, row = q / CPR, c8 = q % CPR;
        mk_ahead[it] = *reinterpret_cast<const h16x8*>(g.mask + (size_t)(m0 + row) * g.ldm + n0 + c8 * 8);
      }
    }
  }

  // ---- epilogue 1: accumulators -> fp32 tile  Tt[wk][m][n]   (C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5))
  float* Tt = reinterpret_cast<float*>(hg_smem);
  {
    float* Tw = Tt + (WK == 1 ? 0 : w * BM * TLD);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = wm * 64 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
          const int n = wn * 64 + j * 32 + l31;
          Tw[m * TLD + n] = acc[i][j][e];
        }
  }
  __syncthreads();

  // ---- epilogue 2: m-major outputs, 8 consecutive n per thread
  static_assert(NT % CPR == 0, "a thread keeps its column chunk over all iterations");
  hg_f32x4 bias0 = hg_f32x4{0.f, 0.f, 0.f, 0.f}, bias1 = bias0;      // q = tid + it * NT: c8 = tid % CPR is the same in every iteration
  if (g.bias) {
    const int gn = n0 + (tid % CPR) * 8;
    bias0 = *reinterpret_cast<const hg_f32x4*>(g.bias + gn); bias1 = *reinterpret_cast<const hg_f32x4*>(g.bias + gn + 4);
  }
  hg_f32x4 seed0 = hg_f32x4{0.f, 0.f, 0.f, 0.f}, seed1 = seed0;
  if (g.seed_w) {
    const int gn = n0 + (tid % CPR) * 8;
    seed0 = *reinterpret_cast<const hg_f32x4*>(g.seed_w + gn); seed1 = *reinterpret_cast<const hg_f32x4*>(g.seed_w + gn + 4);
  }
  float sq = 0.f;                           // sum of squares of the fp32 values this thread writes (clip norm)
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int q = tid + it * NT;
    const int row = q / CPR, c8 = q % CPR;
    float v[8];
    {
      const hg_f32x4* p = reinterpret_cast<const hg_f32x4*>(Tt + row * TLD + c8 * 8);
      hg_f32x4 x0 = p[0], x1 = p[1];
      if constexpr (WK > 1) {
        const hg_f32x4* p1 = reinterpret_cast<const hg_f32x4*>(Tt + (BM + row) * TLD + c8 * 8);
        const hg_f32x4* p2 = reinterpret_cast<const hg_f32x4*>(Tt + (2 * BM + row) * TLD + c8 * 8);
        const hg_f32x4* p3 = reinterpret_cast<const hg_f32x4*>(Tt + (3 * BM + row) * TLD + c8 * 8);
        x0 = (x0 + p1[0]) + (p2[0] + p3[0]);
        x1 = (x1 + p1[1]) + (p2[1] + p3[1]);
      }
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
    const int gm = m0 + row, gn = n0 + c8 * 8;
    if (g.bias) {
      v[0] += bias0.x; v[1] += bias0.y; v[2] += bias0.z; v[3] += bias0.w; v[4] += bias1.x; v[5] += bias1.y; v[6] += bias1.z; v[7] += bias1.w;
    }
    if (g.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
    }
    if (g.mask) {
      h16x8 mk;
      if constexpr (kMaskAhead) mk = mk_ahead[it];
      else mk = *reinterpret_cast<const h16x8*>(g.mask + (size_t)gm * g.ldm + gn);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= ((float)mk[e] > 0.f ? 1.0f : 0.01f);
    }
    if (g.C16) {
      h16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (h16)v[e];
      *reinterpret_cast<h16x8*>(g.C16 + (size_t)gm * g.ldc16 + gn) = o;
      if (g.seed_w) {
        const float sw[8] = {seed0.x, seed0.y, seed0.z, seed0.w, seed1.x, seed1.y, seed1.z, seed1.w};
        h16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (h16)(((-sw[e]) * ((float)o[e] > 0.f ? 1.0f : 0.01f)) * g.seed_scale);
        *reinterpret_cast<h16x8*>(g.CS16 + (size_t)gm * g.ldcs16 + gn) = z;
      }
    }
    if (g.C32 && gn < g.n_valid32) {
      const float s = g.scale32;
      float* d = g.C32 + (size_t)gm * g.ldc32 + gn;
      *reinterpret_cast<hg_f32x4*>(d) = hg_f32x4{v[0] * s, v[1] * s, v[2] * s, v[3] * s};
      *reinterpret_cast<hg_f32x4*>(d + 4) = hg_f32x4{v[4] * s, v[5] * s, v[6] * s, v[7] * s};
#pragma unroll
      for (int e = 0; e < 8; ++e) sq = fmaf(v[e] * s, v[e] * s, sq);
    }
#ifdef HG_WITH_CT16
    if (g.CT16) {                            // keep the finished values for the transposed pass
      hg_f32x4* p = reinterpret_cast<hg_f32x4*>(Tt + row * TLD + c8 * 8);
      p[0] = hg_f32x4{v[0], v[1], v[2], v[3]}; p[1] = hg_f32x4{v[4], v[5], v[6], v[7]};
    }
#endif
  }
  if (g.sumsq_partial) {                    // fixed-order block reduction: lanes, then the waves in index order
    sq = wave_sum64(sq);
    __shared__ float s_sq[NW];
    if (lane == 0) s_sq[w] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) t += s_sq[i];
      g.sumsq_partial[bid] = t;
    }
  }
#ifdef HG_WITH_CT16
  // (test build only: the learner keeps no transposed panel — the dgrad / wgrad operands are read reduction-major)
  if (!g.CT16) return;
  __syncthreads();
  // ---- epilogue 3: transposed output  CT16[n][m]: 4 lanes cover 32 consecutive m of one n (64 B)
  constexpr int CPC = BM / 8;               // 8-m chunks per column
  for (int q = tid; q < BN * CPC; q += NT) {
    const int clo = q & 3, n = (q >> 2) % BN, chi = (q >> 2) / BN;
    const int c8 = chi * 4 + clo;
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)Tt[(c8 * 8 + e) * TLD + n];
    *reinterpret_cast<h16x8*>(g.CT16 + (size_t)(n0 + n) * g.ldct16 + m0 + c8 * 8) = o;
  }
#endif
}

// One launch carries up to kHGemmMax problems of the same tile configuration; MODE0 = the operand orientation of
// problem 0, MODE1 = that of every later problem.
__device__ __forceinline__ int hg_select(const HGemmBatch& batch, int blk, int& bid) {
  int sel = 0;
#pragma unroll
  for (int i = 1; i < kHGemmMax; ++i) if (i < batch.n && blk >= batch.tile_end[i - 1]) sel = i;      // wave-uniform
  bid = blk - (sel ? batch.tile_end[sel - 1] : 0);
  return sel;
}
template <int WM, int WN, int MODE0, int MODE1>
__global__ __launch_bounds__((HGCfg<WM, WN>::NT), 1) void hgemm_nt(HGemmBatch batch) {
  int bid;
  const int sel = hg_select(batch, (int)blockIdx.x, bid);
  if constexpr (MODE0 == MODE1) hgemm_body<WM, WN, MODE0>(batch.g[sel], bid);
  else if (sel == 0) hgemm_body<WM, WN, MODE0>(batch.g[0], bid);
  else hgemm_body<WM, WN, MODE1>(batch.g[sel], bid);
}

#ifdef HG_WITH_CT16
// Probe (test build only): k-major problems on 256 x 128 tiles of FOUR waves, 128 x 64 per wave — 25 %
// fewer LDS fragment bytes and DMA bytes per FLOP than the 64 x 64 wave tile.  Measured (profiles/r03_fp16_gemm_tiles.txt):
// 607 TF against 668 TF for the eight-wave 256 x 128 tile on two 4096 x 1024 x 1024 problems, 784 against 805 TF at K = 4096
// (128 x 128 tiles: 833 TF): the plateau is not set by LDS read traffic.  The learner does not launch it.
static __global__ __launch_bounds__(256, 1) void hgemm_nt_tall(HGemmBatch batch) {
  int bid;
  const int sel = hg_select(batch, (int)blockIdx.x, bid);
  hgemm_body<2, 2, 0, 2>(batch.g[sel], bid);
}
#endif

// the instantiations the learner uses: forward (0), dgrad (2 = B reduction-major), wgrad (3 = both), and a small-
// minibatch layer's dgrad + wgrad in one launch (2, 3)
#define HG_FOR_EACH_KERNEL(X) \
  X(2, 2, 0, 0) X(2, 2, 2, 2) X(2, 2, 3, 3) X(1, 1, 0, 0) X(1, 1, 2, 2) X(1, 1, 3, 3) X(1, 1, 2, 3) X(4, 2, 0, 0)

inline hipError_t hgemm_group_db_prepare();
inline hipError_t hgemm_prepare_all() {
  hipError_t e = hgemm_group_db_prepare();
#ifdef HG_WITH_CT16
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hgemm_nt_tall), hipFuncAttributeMaxDynamicSharedMemorySize, HGCfg<2, 2, 2>::LDS_BYTES);
#endif
#define HG_PREP(WM, WN, M0, M1)                                                                                     \
  if (e == hipSuccess)                                                                                               \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hgemm_nt<WM, WN, M0, M1>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            HGCfg<WM, WN>::LDS_BYTES);
  HG_FOR_EACH_KERNEL(HG_PREP)
#undef HG_PREP
  return e;
}

// Tile choice: 256x128 (8 waves) when THAT fills the chip (two 4096-row problems in one launch), 128x128 when that
// does, else 64x64 with in-workgroup split-K.  force: 0 auto, 1 128x128, 2 64x64, 3 256x128 on eight waves,
// 4 256x128 on four waves (128x64 per wave).
inline bool hgemm_big_ok(const HGemm& g) { return (g.M % 128 == 0) && (g.N % 128 == 0); }
inline bool hgemm_huge_ok(const HGemm& g) { return (g.M % 256 == 0) && (g.N % 128 == 0) && !g.ta && !g.tb; }
inline long hgemm_tiles(const HGemm& g, bool big) { return big ? (long)(g.M / 128) * (g.N / 128) : (long)(g.M / 64) * (g.N / 64); }
inline int hgemm_mode(const HGemm& g) { return (g.ta ? 1 : 0) | (g.tb ? 2 : 0); }

// fills b (problems, tile ranges) and returns the tile configuration in wm / wn; hipErrorInvalidValue if the
// problems do not fit one
inline hipError_t hgemm_plan(const HGemm* gs, int n, int force, HGemmBatch& b, int& wm, int& wn, long& blocks) {
  if (n < 1 || n > kHGemmMax) return hipErrorInvalidValue;
  bool big_ok = true, huge_ok = true; long tiles_big = 0, tiles_huge = 0;
  for (int i = 0; i < n; ++i) {
    big_ok = big_ok && hgemm_big_ok(gs[i]); huge_ok = huge_ok && hgemm_huge_ok(gs[i]);
    if (gs[i].K % 64 || gs[i].K < 64) return hipErrorInvalidValue;
  }
  if (big_ok) for (int i = 0; i < n; ++i) tiles_big += hgemm_tiles(gs[i], true);
  if (huge_ok) for (int i = 0; i < n; ++i) tiles_huge += (long)(gs[i].M / 256) * (gs[i].N / 128);
  const bool huge = force == 3 || force == 4 || (force == 0 && huge_ok && tiles_huge >= 192);
  const bool big = !huge && (force == 1 || (force == 0 && big_ok && tiles_big >= 192));
  if (huge && !huge_ok) return hipErrorInvalidValue;
  if (big && !big_ok) return hipErrorInvalidValue;
  b = HGemmBatch{};
  b.n = n;
  blocks = 0;
  for (int i = 0; i < n; ++i) {
    b.g[i] = gs[i];
    if (huge) blocks += (long)(gs[i].M / 256) * (gs[i].N / 128);
    else if (big) blocks += hgemm_tiles(gs[i], true);
    else { if (gs[i].M % 64 || gs[i].N % 64 || gs[i].K % 128) return hipErrorInvalidValue; blocks += hgemm_tiles(gs[i], false); }
    b.tile_end[i] = (int)blocks;
  }
  wm = huge ? 4 : (big ? 2 : 1); wn = huge ? 2 : (big ? 2 : 1);
  return hipSuccess;
}

inline hipError_t hgemm_launch_batch(const HGemm* gs, int n, hipStream_t st, int force = 0, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
  HGemmBatch b; int wm, wn; long blocks;
  hipError_t e = hgemm_plan(gs, n, force, b, wm, wn, blocks);
  if (e != hipSuccess) return e;
  const int m0 = hgemm_mode(gs[0]), m1 = n > 1 ? hgemm_mode(gs[1]) : m0;
  for (int i = 2; i < n; ++i) if (hgemm_mode(gs[i]) != m1) return hipErrorInvalidValue;
  if (force == 4) {
#ifdef HG_WITH_CT16
    if (m0 != 0 || m1 != 0) return hipErrorInvalidValue;
    if (t0) hipExtLaunchKernelGGL(hgemm_nt_tall, dim3((unsigned)blocks), dim3(256), (HGCfg<2, 2, 2>::LDS_BYTES), st, t0, t1, 0, b);
    else hipLaunchKernelGGL(hgemm_nt_tall, dim3((unsigned)blocks), dim3(256), (HGCfg<2, 2, 2>::LDS_BYTES), st, b);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;
#endif
  }
  bool launched = false;
#define HG_TRY(WM, WN, M0, M1)                                                                                                          \
  if (!launched && wm == WM && wn == WN && m0 == M0 && m1 == M1) {                                                                      \
    launched = true;                                                                                                                    \
    if (t0) hipExtLaunchKernelGGL((hgemm_nt<WM, WN, M0, M1>), dim3((unsigned)blocks), dim3(HGCfg<WM, WN>::NT), (HGCfg<WM, WN>::LDS_BYTES), st, t0, t1, 0, b); \
    else hipLaunchKernelGGL((hgemm_nt<WM, WN, M0, M1>), dim3((unsigned)blocks), dim3(HGCfg<WM, WN>::NT), (HGCfg<WM, WN>::LDS_BYTES), st, b);         \
  }
  HG_FOR_EACH_KERNEL(HG_TRY)
#undef HG_TRY
  if (!launched) return hipErrorInvalidValue;          // an orientation pair no kernel was built for
  return hipGetLastError();
}
inline hipError_t hgemm_launch(const HGemm& g, hipStream_t st, int force = 0, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
  return hgemm_launch_batch(&g, 1, st, force, t0, t1);
}
// would a stand-alone launch of g use the 64x64 split-K tile?
inline bool hgemm_uses_small_tile(const HGemm& g) { return !(hgemm_big_ok(g) && hgemm_tiles(g, true) >= 192); }

// ---------------------------------------------------------------------------------------------
// fp32 [rows][ld_src] -> fp16 [rows][ld16] (+ transposed fp16 [ld16][ldT]) glue: minibatch panels,
// head gradients (with the loss scale) and the per-update fp16 weight copies.
struct Cvt16 {
  const float* src; int ld_src; int rows; int cols;   // cols = valid source columns (<= ld_src)
  h16* dst; int ld16;                                 // [rows][ld16], columns >= cols written as 0 (may be null)
  h16* dstT; int ldT;                                 // [ld16][ldT] transposed (test build only; null in the learner)
  float scale;
  int tiles_r, tiles_c, tile_base;
};
struct Cvt16Batch { Cvt16 d[8]; int n; };

template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_cvt16(Cvt16Batch b) {
  __shared__ float t[64][65];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.d[j + 1].tile_base) ++j;
  const Cvt16& d = b.d[j];
  const int tl = blockIdx.x - d.tile_base;
  const int r0 = (tl / d.tiles_c) * 64, c0 = (tl % d.tiles_c) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int row = r0 + r, col = c0 + tx;
    float v = 0.f;
    if (row < d.rows && col < d.cols) v = d.src[(size_t)row * d.ld_src + col] * d.scale;
    t[r][tx] = v;
    if (d.dst && row < d.rows) d.dst[(size_t)row * d.ld16 + col] = (h16)v;
  }
#ifdef HG_WITH_CT16
  if (!d.dstT) return;
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int row = r0 + tx;
    if (row < d.ldT) d.dstT[(size_t)(c0 + c) * d.ldT + row] = (h16)(row < d.rows ? t[tx][c] : 0.f);
  }
#endif
}

inline void cvt16_add(Cvt16Batch& b, const float* src, int ld_src, int rows, int cols, h16* dst, int ld16, float scale, h16* dstT = nullptr, int ldT = 0) {
  Cvt16& d = b.d[b.n];
  d.src = src; d.ld_src = ld_src; d.rows = rows; d.cols = cols; d.dst = dst; d.ld16 = ld16; d.dstT = dstT; d.ldT = ldT; d.scale = scale;
  d.tiles_r = (rows + 63) / 64; d.tiles_c = ld16 / 64;
  d.tile_base = b.n ? b.d[b.n - 1].tile_base + b.d[b.n - 1].tiles_r * b.d[b.n - 1].tiles_c : 0;
  b.n += 1;
}
inline hipError_t cvt16_launch(const Cvt16Batch& b, hipStream_t st) {
  if (!b.n) return hipSuccess;
  const Cvt16& l = b.d[b.n - 1];
  hipLaunchKernelGGL(k_cvt16<0>, dim3(l.tile_base + l.tiles_r * l.tiles_c), dim3(256), 0, st, b);
  return hipGetLastError();
}

#ifdef HG_WITH_CT16
// bias gradients from TRANSPOSED panels (round-1 form, test build only): db_l[n] = scale * sum_b dYT_l[n][b]
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_db16(Db16Batch b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.d[j + 1].row_base) ++j;
  const Db16& d = b.d[j];
  const int n = blockIdx.x - d.row_base;
  const h16* p = d.dyt + (size_t)n * d.ld;
  float acc = 0.f;
  for (int i = threadIdx.x * 8; i < d.rows; i += 256 * 8) {
    const h16x8 v = *reinterpret_cast<const h16x8*>(p + i);
    acc += (((float)v[0] + (float)v[1]) + ((float)v[2] + (float)v[3])) + (((float)v[4] + (float)v[5]) + ((float)v[6] + (float)v[7]));
  }
  acc = wave_sum64(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) d.db[n] = ((s[0] + s[1]) + (s[2] + s[3])) * b.scale;
}
#endif

// the same from the batch-major panels dY [rows][ld] (no transposed copy exists): a block owns 64 columns of one layer,
// 8 lanes x 16 B cover them, 32 row groups stride the rows; fixed-order LDS reduction over the row groups.
// (Db16::dyt = the panel, ld = its row stride, row_base counts 64-column blocks.)
__device__ __forceinline__ void db16_cols_block(const Db16Batch& b, int blk, float* scratch /* 32 x 65 floats of LDS */) {
  float (*sred)[65] = reinterpret_cast<float (*)[65]>(scratch);
  int j = 0;
  while (j + 1 < b.n && blk >= b.d[j + 1].row_base) ++j;
  const Db16& d = b.d[j];
  const int col0 = (blk - d.row_base) * 64;
  const int c8 = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const h16* p = d.dyt + col0 + c8 * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int r = rg;
  for (; r + 96 < d.rows; r += 128) {        // four independent 16-B loads in flight
    const h16x8 v0 = *reinterpret_cast<const h16x8*>(p + (size_t)r * d.ld), v1 = *reinterpret_cast<const h16x8*>(p + (size_t)(r + 32) * d.ld);
    const h16x8 v2 = *reinterpret_cast<const h16x8*>(p + (size_t)(r + 64) * d.ld), v3 = *reinterpret_cast<const h16x8*>(p + (size_t)(r + 96) * d.ld);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
  }
  for (; r < d.rows; r += 32) {
    const h16x8 v = *reinterpret_cast<const h16x8*>(p + (size_t)r * d.ld);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sred[rg][c8 * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += sred[g][threadIdx.x];
    s *= b.scale;
    d.db[col0 + threadIdx.x] = s;
    if (b.sumsq_partial) {                   // one wave: butterfly, lane 0 stores
      float q = s * s;
      q = wave_sum64(q);
      if (threadIdx.x == 0) b.sumsq_partial[blk] = q;
    }
  }
}
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_db16_cols(Db16Batch b) {
  __shared__ float sred[32 * 65];
  db16_cols_block(b, (int)blockIdx.x, sred);
}

// ALL wgrads of a net in one launch, with the bias-gradient column sums of every layer as extra workgroups.
// Alone, a layer's wgrad (N_out x K_in = 1024 x 1024, reduction = the minibatch) is 64 tiles of 128 x 128 — a quarter
// of the chip — or 256 of 64 x 64 at four times the operand bytes per FLOP (455 TF at 4096 rows, r02 profile);
// once the dgrad chain has produced every dZ panel the L wgrads are independent, and together they are
// 3 x 64 + 8 = 200 tiles of 128 x 128 (+ 48 column-sum workgroups) at 4 x 1024: one round on 256 CUs.
// Round 6: the head's own gradients as column-sum workgroups of the same launch.  dWh[j][k] = sum_m dy[m][j] X[m][k] is a WEIGHTED
// column sum of the tower top (the weights: the head diffs of row m), db_h[j] = sum_m dy[m][j]: the streaming pattern of
// db16_cols_block — a block owns 64 columns, 8 lanes x 16 B cover them, 32 row groups stride the rows, fixed-order LDS reduction
// over the row groups — with NH accumulator sets.  H / 64 short blocks (+ db in block 0) instead of the head-backward kernels'
// slab / ticket tails (k_head_bwd) or a reduction launch of their own (k_head_wred); the head-backward kernel then only produces
// dZ, and Step(1)'s needs no launch at all (k_head_q_train writes the tower-top gradient).
// (First form, measured: the fp32 path's HeadWgradRider blocks — 8 columns x every row, H / 8 = 128 blocks — as riders of this
// launch: with one 128-KiB workgroup per CU they queue behind the 200 wgrad tiles and gave back what the removed launch had saved:
// 0.2626 against 0.263 ms per update at 512 rows.)
struct HeadWsum {
  const float* dy; int lddy;                // head diffs [rows][lddy] (critic: dq, lddy 1; actor: the post-invert diffs, lddy 16)
  const h16* X16; int H, rows;              // tower top [rows][H]
  float* dW; float* db; float* partial;     // [NH][H], [NH], one sum-of-squares slot per block
  int nh;                                   // 1 / 10 (0: none)
  int blocks;                               // H / 64
};
// (Minibatches below 1024 rows only.  At 4096 rows the blocks extend the 43-us grouped wgrad launch they ride in — one 128-KiB workgroup
// per CU: what does not fit beside the 200 tiles queues — by 9 us (critic) / 14 us (actor heads), more than the head-backward launches
// they replace cost; 64- and 16-column blocks, either order in the grid, 2 to 8 rows in flight: same-box A/B 0.607-0.614 against
// 0.597-0.605 ms per update without them, profiles/r06_fp16_merges_ab.txt.)
template <int NH, int CW = 64>
__device__ __forceinline__ void head_wsum_block(const HeadWsum& r, int blk, float* scratch /* < 100 KB of LDS: NH x (2048 / CW) x (CW + 1) + NH x 256 floats */) {
  constexpr int LPR = CW / 8, RG = 256 / LPR;      // lanes per row, row groups
  const int col0 = blk * CW;
  const int tid = threadIdx.x, c8 = tid % LPR, rg = tid / LPR;
  const h16* p = r.X16 + col0 + c8 * 8;
  float acc[NH][8];
#pragma unroll
  for (int j = 0; j < NH; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
  auto weights = [&](int row, float (&w)[NH]) {
    if constexpr (NH == 1) w[0] = r.dy[(size_t)row * r.lddy];
    else {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4* q = reinterpret_cast<const f4*>(r.dy + (size_t)row * r.lddy);      // (lddy = 16: 64-B rows)
#pragma unroll
      for (int j4 = 0; j4 < (NH + 3) / 4; ++j4) {
        const f4 v = q[j4];
#pragma unroll
        for (int c = 0; c < 4; ++c) if (j4 * 4 + c < NH) w[j4 * 4 + c] = v[c];
      }
    }
  };
  // U rows in flight per thread
  constexpr int U = NH == 1 ? 8 : 4;
  int row = rg;
  for (; row + RG * (U - 1) < r.rows; row += RG * U) {
    h16x8 v[U]; float w[U][NH];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = *reinterpret_cast<const h16x8*>(p + (size_t)(row + RG * u) * r.H); weights(row + RG * u, w[u]); }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(w[u][j], (float)v[u][e], acc[j][e]);
  }
  for (; row < r.rows; row += RG) {
    const h16x8 v0 = *reinterpret_cast<const h16x8*>(p + (size_t)row * r.H);
    float w0[NH];
    weights(row, w0);
#pragma unroll
    for (int j = 0; j < NH; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(w0[j], (float)v0[e], acc[j][e]);
  }
  // Reduction over the row groups, all NH sets at once (three barriers): every set to LDS, PARTS = 256 / CW partial sums per (head,
  // column) over RG / PARTS consecutive row groups each, then the partials in index order — a fixed order.
  // (First form: per head, CW threads summing all RG row groups serially — 10 x 128 dependent LDS reads at 4096 rows = 18 us that
  // extended the 40-us launch this block rides in.)
  constexpr int PARTS = 256 / CW, PER = RG / PARTS, LD = CW + 1;
  float* S = scratch;                                  // [NH][RG][LD]
  float* S2 = scratch + NH * RG * LD;                  // [NH][PARTS][CW]
#pragma unroll
  for (int j = 0; j < NH; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) S[(j * RG + rg) * LD + c8 * 8 + e] = acc[j][e];
  __syncthreads();
  {
    const int col = tid % CW, part = tid / CW;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < PER; ++g) t += S[(j * RG + part * PER + g) * LD + col];
      S2[(j * PARTS + part) * CW + col] = t;
    }
  }
  __syncthreads();
  float ssq = 0.f;
  for (int o = tid; o < NH * CW; o += 256) {
    const int j = o / CW, c = o % CW;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) t += S2[(j * PARTS + q) * CW + c];
    r.dW[(size_t)j * r.H + col0 + c] = t;
    ssq = fmaf(t, t, ssq);
  }
  if (blk == 0) {                            // the head's bias gradient: rows strided over the threads, fixed-order reduction
    float sb[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) sb[j] = 0.f;
    for (int m = tid; m < r.rows; m += 256) {
      float w[NH];
      weights(m, w);
#pragma unroll
      for (int j = 0; j < NH; ++j) sb[j] += w[j];
    }
    __syncthreads();                         // (S is free again)
#pragma unroll
    for (int j = 0; j < NH; ++j) { const float t = wave_sum64(sb[j]); if ((tid & 63) == 0) S[(tid >> 6) * 16 + j] = t; }
    __syncthreads();
    if (tid < NH) {
      const float v = (S[0 * 16 + tid] + S[1 * 16 + tid]) + (S[2 * 16 + tid] + S[3 * 16 + tid]);
      r.db[tid] = v;
      ssq = fmaf(v, v, ssq);
    }
  }
  if (r.partial != nullptr) {                // one slot per block: the four waves' sums in fixed order
    ssq = wave_sum64(ssq);
    __syncthreads();
    if ((tid & 63) == 0) S2[tid >> 6] = ssq;
    __syncthreads();
    if (tid == 0) r.partial[blk] = (S2[0] + S2[1]) + (S2[2] + S2[3]);
  }
}
// ... and, for data-parallel learners, the tails block (TailsArgs).  Grid: tiles, head blocks, column-sum blocks, tails.
template <int WM, int WN>
__global__ __launch_bounds__((HGCfg<WM, WN>::NT), 1) void hgemm_group_db(HGemmBatch batch, Db16Batch db, int db_blocks, HeadWsum head, TailsArgs tails) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char hg_smem[];
  const int nt = batch.tile_end[batch.n - 1];
  int b = (int)blockIdx.x;
  if (b < nt) {
    int bid;
    const int sel = hg_select(batch, b, bid);
    hgemm_body<WM, WN, 3>(batch.g[sel], bid);
    return;
  }
  b -= nt;
  const int nhb = head.nh != 0 ? head.blocks : 0;     // (the head blocks before the shorter column-sum blocks: with one workgroup per CU what does not fit beside the tiles queues)
  if (b < nhb) {
    float* sc = reinterpret_cast<float*>(hg_smem);
    if (head.nh == 1) head_wsum_block<1>(head, b, sc); else head_wsum_block<10>(head, b, sc);
    return;
  }
  b -= nhb;
  if (b < db_blocks) { db16_cols_block(db, b, reinterpret_cast<float*>(hg_smem)); return; }
  if (tails.on) tails_block(tails, reinterpret_cast<float*>(hg_smem), reinterpret_cast<double*>(hg_smem + 64));
}
inline hipError_t hgemm_group_db_prepare() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hgemm_group_db<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, HGCfg<1, 1>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hgemm_group_db<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, HGCfg<2, 2>::LDS_BYTES);
  return e;
}
// gs: n reduction-major wgrads (mode 3); db_blocks = 64-column blocks of db (0: none).  big: 128 x 128 tiles
// (every M, N a multiple of 128), else 64 x 64 split-K tiles.
inline hipError_t hgemm_group_db_launch(const HGemm* gs, int n, bool big, const Db16Batch& db, int db_blocks, hipStream_t st,
                                        hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, const HeadWsum* head_in = nullptr, const TailsArgs* tails_in = nullptr) {
  for (int i = 0; i < n; ++i) if (hgemm_mode(gs[i]) != 3) return hipErrorInvalidValue;
  HGemmBatch b; int wm, wn; long blocks;
  hipError_t e = hgemm_plan(gs, n, big ? 1 : 2, b, wm, wn, blocks);
  if (e != hipSuccess) return e;
  HeadWsum head{}; TailsArgs tails{};
  if (head_in != nullptr) head = *head_in;
  if (head.nh != 0 && ((head.nh != 1 && head.nh != 10) || head.blocks * 64 != head.H)) return hipErrorInvalidValue;
  if (tails_in != nullptr) { tails = *tails_in; tails.on = 1; }
  const unsigned grid = (unsigned)(blocks + db_blocks + (head.nh ? head.blocks : 0) + (tails.on ? 1 : 0));
  if (big) {
    if (t0) hipExtLaunchKernelGGL((hgemm_group_db<2, 2>), dim3(grid), dim3(256), (HGCfg<2, 2>::LDS_BYTES), st, t0, t1, 0, b, db, db_blocks, head, tails);
    else hipLaunchKernelGGL((hgemm_group_db<2, 2>), dim3(grid), dim3(256), (HGCfg<2, 2>::LDS_BYTES), st, b, db, db_blocks, head, tails);
  } else {
    if (t0) hipExtLaunchKernelGGL((hgemm_group_db<1, 1>), dim3(grid), dim3(256), (HGCfg<1, 1>::LDS_BYTES), st, t0, t1, 0, b, db, db_blocks, head, tails);
    else hipLaunchKernelGGL((hgemm_group_db<1, 1>), dim3(grid), dim3(256), (HGCfg<1, 1>::LDS_BYTES), st, b, db, db_blocks, head, tails);
  }
  return hipGetLastError();
}

}  // namespace dqnhip
