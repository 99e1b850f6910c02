// learner.hip — the device-resident actor-critic learner behind include/dqnhip.h.
//
// Host-side orchestration of DQN::UpdateActorCritic (reference src/dqn.cpp:828-972)
// as a fixed sequence of gfx950 kernels on one HIP stream with no host sync:
// nothing crosses PCIe per update except (optionally) B sampled indices in and
// two floats out.  See DESIGN.md for the data layout and the kernel list.
// This translation unit: the update itself.  The data-parallel exchange is in learner_dp.hip, acting / replay / parameters /
// sharing / introspection in learner_io.hip, the env front-end's host side in learner_env.hip (learner_internal.hip.h).
#include "learner_internal.hip.h"

using namespace dqnhip;
using namespace dqnhip_host;

namespace dqnhip_host {

thread_local std::string g_err;

void layout_init(NetLayout& l, int in_dim, const dqnhip_config& c, bool actor) {
  l.L = c.num_hidden; l.in_dim = in_dim; l.NH = actor ? kNO : 1;
  // fp16 mode: 128-wide first panel so that the fp16 weight arena mirrors this one offset for offset
  l.dims[0] = in_dim; l.kp[0] = round_up(in_dim, c.precision == DQNHIP_FP16 ? 128 : 64);
  for (int i = 0; i < l.L; ++i) { l.dims[i + 1] = c.hidden[i]; l.kp[i + 1] = c.hidden[i]; }
  size_t off = 0, dense = 0;
  int part = 0;
  for (int i = 0; i < l.L; ++i) {
    l.w_off[i] = off; off += (size_t)l.dims[i + 1] * l.kp[i];
    l.b_off[i] = off; off += round_up(l.dims[i + 1], 64);
    dense += (size_t)l.dims[i + 1] * l.dims[i] + l.dims[i + 1];
    // one slot per wgrad tile; the first layer may run the 16-output tiles of wgrad_narrow_body
    l.part_off[i] = part; part += (l.kp[i] / 64) * (l.dims[i + 1] / (i == 0 ? 16 : 64));
  }
  const int H = l.dims[l.L];
  l.hw_off = off; off += round_up_z((size_t)l.NH * H, 64);
  l.hb_off = off; off += 64;
  dense += (size_t)l.NH * H + l.NH;
  l.part_off[l.L] = part; part += std::max((H / 64) * l.NH, H / kRiderCW);    // k_head_bwd uses the first H/64, head_wgrad_rider H/8, k_head_wred one per (head, 64 columns)
  // fp16 learner: the bias gradients come from their own workgroups (k_db16_cols, one per 64 columns): their slots
  l.part_db = part;
  for (int i = 0; i < l.L; ++i) part += l.dims[i + 1] / 64;
  l.arena = round_up_z(off, 64);
  l.dense = dense;
  l.n_part = part;
}

const char* const kFamily[] = {"gemm_fwd_lds_4x2", "gemm_dgrad", "gemm_wgrad", "adam", "gemm_bwd_pair", "gemm_fwd_lds_2x2", "gemm_fwd_direct",
                               "hgemm_fwd", "hgemm_dgrad", "hgemm_wgrad"};

// ---- dense (Caffe order) <-> internal arena ----------------------------------
void dense_to_arena(const NetLayout& l, const float* dense, std::vector<float>& arena) {
  arena.assign(l.arena, 0.0f);
  size_t d = 0;
  for (int i = 0; i < l.L; ++i) {
    const int N = l.dims[i + 1], K = l.dims[i], KP = l.kp[i];
    for (int n = 0; n < N; ++n) memcpy(&arena[l.w_off[i] + (size_t)n * KP], dense + d + (size_t)n * K, K * sizeof(float));
    d += (size_t)N * K;
    memcpy(&arena[l.b_off[i]], dense + d, N * sizeof(float)); d += N;
  }
  const int Hh = l.dims[l.L];
  if (l.NH == kNO) {  // action_layer.W[4,H] .b[4] actionpara_layer.W[6,H] .b[6]
    memcpy(&arena[l.hw_off], dense + d, (size_t)kNA * Hh * sizeof(float)); d += (size_t)kNA * Hh;
    memcpy(&arena[l.hb_off], dense + d, kNA * sizeof(float)); d += kNA;
    memcpy(&arena[l.hw_off + (size_t)kNA * Hh], dense + d, (size_t)kNP * Hh * sizeof(float)); d += (size_t)kNP * Hh;
    memcpy(&arena[l.hb_off + kNA], dense + d, kNP * sizeof(float)); d += kNP;
  } else {
    memcpy(&arena[l.hw_off], dense + d, (size_t)Hh * sizeof(float)); d += Hh;
    arena[l.hb_off] = dense[d]; d += 1;
  }
}
void arena_to_dense(const NetLayout& l, const std::vector<float>& arena, float* dense) {
  size_t d = 0;
  for (int i = 0; i < l.L; ++i) {
    const int N = l.dims[i + 1], K = l.dims[i], KP = l.kp[i];
    for (int n = 0; n < N; ++n) memcpy(dense + d + (size_t)n * K, &arena[l.w_off[i] + (size_t)n * KP], K * sizeof(float));
    d += (size_t)N * K;
    memcpy(dense + d, &arena[l.b_off[i]], N * sizeof(float)); d += N;
  }
  const int Hh = l.dims[l.L];
  if (l.NH == kNO) {
    memcpy(dense + d, &arena[l.hw_off], (size_t)kNA * Hh * sizeof(float)); d += (size_t)kNA * Hh;
    memcpy(dense + d, &arena[l.hb_off], kNA * sizeof(float)); d += kNA;
    memcpy(dense + d, &arena[l.hw_off + (size_t)kNA * Hh], (size_t)kNP * Hh * sizeof(float)); d += (size_t)kNP * Hh;
    memcpy(dense + d, &arena[l.hb_off + kNA], kNP * sizeof(float)); d += kNP;
  } else {
    memcpy(dense + d, &arena[l.hw_off], (size_t)Hh * sizeof(float)); d += Hh;
    dense[d] = arena[l.hb_off]; d += 1;
  }
}


int validate(const dqnhip_config* c) {
  if (!c) return fail("config is null");
  if (c->struct_size != (int32_t)sizeof(dqnhip_config)) return fail("dqnhip_config.struct_size %d != %zu (ABI mismatch)", c->struct_size, sizeof(dqnhip_config));
  if (c->minibatch <= 0 || c->minibatch % 32) return fail("minibatch must be a positive multiple of 32 (got %d)", c->minibatch);
  if (c->state_size < 1) return fail("state_size must be >= 1");
  if (c->num_hidden < 1 || c->num_hidden > kMaxL) return fail("num_hidden out of range");
  for (int i = 0; i < c->num_hidden; ++i)
    if (c->hidden[i] <= 0 || c->hidden[i] % 64) return fail("hidden[%d]=%d must be a positive multiple of 64", i, c->hidden[i]);
  if (c->replay_capacity < 2) return fail("replay_capacity must be >= 2");
  if (c->soft_update_freq < 1) return fail("soft_update_freq must be >= 1");
  if (c->dp_world < 1 || c->dp_rank < 0 || c->dp_rank >= c->dp_world) return fail("bad dp_world/dp_rank");
  if (c->precision != DQNHIP_FP32 && c->precision != DQNHIP_FP16) return fail("precision must be DQNHIP_FP32 or DQNHIP_FP16");
  if (c->precision == DQNHIP_FP16) {
    if (c->minibatch % 128) return fail("fp16 mode: minibatch must be a multiple of 128 (got %d)", c->minibatch);
    for (int i = 0; i < c->num_hidden; ++i)
      if (c->hidden[i] % 128) return fail("fp16 mode: hidden[%d]=%d must be a multiple of 128", i, c->hidden[i]);
  }
  return 0;
}


// ---- forward / backward building blocks ----------------------------------------

// One tower layer forward for up to kMaxGroup passes of identical shape.
int layer_forward(H* h, hipStream_t st, const FwdPass* passes, int n, int rows, int i) {
  const NetLayout& l = *passes[0].l;
  GemmBatch b{}; b.n = n;
  for (int j = 0; j < n; ++j) {
    GemmProblem& p = b.prob[j];
    p.P = wat(h, passes[j].net, l.w_off[i]); p.ldp = l.kp[i];
    p.Q = passes[j].act[i]; p.ldq = l.kp[i];
    p.C = passes[j].act[i + 1]; p.ldc = l.kp[i + 1];
    p.Pdim = l.dims[i + 1]; p.Qdim = rows; p.Kred = l.kp[i];
    p.bias = wat(h, passes[j].net, l.b_off[i]); p.relu = 1;
    if (i == l.L - 1 && passes[j].seed_w != nullptr) { p.seed_w = passes[j].seed_w; p.C2 = passes[j].seed_out; }
    if (i == l.L - 1 && passes[j].dot_w != nullptr) { p.dot_w = passes[j].dot_w; p.dot_out = passes[j].dot_out; }
  }
  // K >= 512 and K % 256 == 0: full-line loads + wave-private LDS transpose; else (first
  // layer, narrow towers) the plain direct kernel.  One problem: 32x32 tiles (256 workgroups
  // for a 256x1024 layer); grouped: 64x32.
  const bool lds_ok = (l.kp[i] >= 512) && (l.kp[i] % 256 == 0);
  ScopedTiming t(h, !lds_ok ? 6 : (n == 1 ? 5 : 0), st);
  // acting-time batches (<= 128 rows, e.g. 64 env workers): 16x16 tiles so that one layer still spreads
  // over 256 workgroups (64 rows x 1024 outputs = 256 tiles) instead of 64
  if (n == 1 && rows <= 128 && lds_ok) HIPCHK((fwd_lds_launch<1, 1, true>(b, st)));
  else if (n == 1 && rows >= 512 && lds_ok && l.dims[i + 1] % 64 == 0) HIPCHK((fwd_lds_launch<4, 2, true>(b, st)));   // enough rows to fill the chip with 64x32 tiles (fewer bytes per FLOP)
  else if (n == 1) { if (lds_ok) HIPCHK((fwd_lds_launch<2, 2, true>(b, st))); else HIPCHK((fwd_direct_launch<2, 2>(b, st))); }
  else {
    // grouped launches: 64x32 tiles when they still give the chip something to do; small minibatches / narrow layers
    // (the reference's defaults: 32 rows into 512 outputs = 16 such tiles for two problems, one long chain each) take
    // 32x32 or 16x16 tiles — same K split over the four waves, same reduction order, more workgroups
    const long P = l.dims[i + 1];
    const long t42 = (long)n * (P / 64) * (rows / 32), t22 = (long)n * (P / 32) * (rows / 32);
    if (lds_ok) {
      if (t42 >= 192) HIPCHK((fwd_lds_launch<4, 2, true, 1>(b, st)));   // one LDS image per wave (48 KiB): both tiles of a CU resident at once — same-box A/B +0.6 %
      else if (t22 >= 128 || rows > 128 || rows % 16) HIPCHK((fwd_lds_launch<2, 2, true>(b, st)));
      else HIPCHK((fwd_lds_launch<1, 1, true>(b, st)));
    } else {
      if (t42 >= 64) HIPCHK((fwd_direct_launch<4, 2>(b, st)));
      else HIPCHK((fwd_direct_launch<2, 2>(b, st)));
    }
  }
  return 0;
}
int tower_forward(H* h, hipStream_t st, const FwdPass* passes, int n, int rows, int first_layer) {
  for (int i = first_layer; i < passes[0].l->L; ++i) RC(layer_forward(h, st, passes, n, rows, i));
  return 0;
}

// Tower backward from dZ[L] (gradient wrt the last tower pre-activation) down, one launch per layer on `st`.
// want_w: produce dW/db (+sumsq partials) into garena; input_grad: also dZ[0].
// in_lo / in_hi: when only these input columns of dZ[0] are consumed (the critic's action columns), the
// first layer's dgrad computes just the 16-column tiles that cover them.
// RCCL sum all-reduce of one slice of a gradient arena on the communication stream, ordered after
// everything enqueued on `st` so far (per-layer bucketing; defined with dqnhip_dp_*)
// does layer i's backward (dgrad + wgrad) take the side-by-side pair launch (small minibatches / narrow layers)?
inline bool bwd_layer_is_pair(const NetLayout& l, int i, int rows) {
  const long tiles = (long)(l.kp[i] / 64) * (rows / 16) + (long)(l.kp[i] / 64) * (l.dims[i + 1] / 64);
  return tiles <= 256 && rows % 16 == 0 && l.kp[i] % 64 == 0 && l.dims[i + 1] % 64 == 0;
}
// may the head's weight / bias gradients ride in the first tower layer's wgrad launch (gemm_wgrad_narrow_rider: the last
// launch of a net's backward, input_grad == false)?
inline bool head_wgrad_can_ride(const NetLayout& l, int rows) {
  const int NH = l.NH, H = l.dims[l.L];
  return H % kRiderCW == 0 && (size_t)(rows * NH + 256 * NH) * sizeof(float) <= (size_t)64 * 1024;
}
// does a weights-wanted, no-input-gradient backward of this tower take the SHIFTED schedule (see tower_backward)?
inline bool bwd_is_shifted(const H* h, const NetLayout& l, int rows) {
  bool shifted = l.L >= 2 && rows % 16 == 0 && !(h->cfg.tuning_flags & DQNHIP_TUNE_BWD_UNSHIFTED);
  for (int i = 1; i < l.L && shifted; ++i) shifted = !bwd_layer_is_pair(l, i, rows) && l.kp[i] % 64 == 0 && l.dims[i + 1] % 64 == 0;
  return shifted;
}
// fuse: the critic's dQ/da pass — the first layer's action-column tiles, the inverting gradients and the actor heads' backward in
// ONE launch (k_dqda_head_bwd) instead of the narrow dgrad launch here and a head-backward launch after it; carries the q rider.
// qtrain (Step(1)'s backward, shifted schedule): the top layer's dgrad launch also does k_head_q_train's work (k_dgrad_qtrain);
// qtrain_seed = the panel U = (-w_h) lrelu'(x_L) the online critic's top forward layer left (the dgrad's dY operand)
int tower_backward(H* h, hipStream_t st, const NetLayout& l, int net, float* garena, float* partial,
                   float** act, float** dZ, int rows, bool want_w, bool input_grad, int in_lo = 0, int in_hi = -1,
                   const HeadWgradRider* rider = nullptr, const QHeadRider* qrider = nullptr, DqdaHeadArgs* fuse = nullptr,
                   const HeadTrainArgs* qtrain = nullptr, const float* qtrain_seed = nullptr, const TailsArgs* tails = nullptr) {
  auto dgrad_of = [&](int i) {             // dZ[i] = (dZ[i+1] . W_i) * lrelu'(act[i])
    GemmProblem p{};
    p.mode = GEMM_DGRAD;
    p.P = wat(h, net, l.w_off[i]); p.ldp = l.kp[i];
    p.Q = dZ[i + 1]; p.ldq = l.kp[i + 1];
    p.C = dZ[i]; p.ldc = l.kp[i];
    p.Pdim = l.kp[i]; p.Qdim = rows; p.Kred = l.dims[i + 1];
    p.mask = i > 0 ? act[i] : nullptr; p.ldm = l.kp[i];
    return p;
  };
  auto wgrad_of = [&](int i) {             // dW_i = dZ[i+1]^T . act[i] ; db_i = colsum(dZ[i+1])
    GemmProblem p{};
    p.mode = GEMM_WGRAD;
    p.P = act[i]; p.ldp = l.kp[i];
    p.Q = dZ[i + 1]; p.ldq = l.kp[i + 1];
    p.C = garena + l.w_off[i]; p.ldc = l.kp[i];
    p.Pdim = l.kp[i]; p.Qdim = l.dims[i + 1]; p.Kred = rows;
    p.db = garena + l.b_off[i];
    p.partial = partial ? partial + l.part_off[i] : nullptr;
    return p;
  };
  // reduction width (the layer's outputs) wide enough: dY through the LDS transpose, scheduled form
  auto lds_ok_of = [&](int i) { return l.dims[i + 1] >= 512 && l.dims[i + 1] % 256 == 0; };
  auto layer_slice = [&](int i) { return (i + 1 < l.L ? l.w_off[i + 1] : l.hw_off) - l.w_off[i]; };
  // SHIFTED schedule (weights wanted, no input gradient, every layer above the first on the one-workgroup-type form):
  //   dgrad(L-1) | wgrad(L-1) + dgrad(L-2) | ... | wgrad(2) + dgrad(1) | wgrad(1) + wgrad(0) + the head's riders
  // instead of  wgrad(i) + dgrad(i) per layer and a last launch with the first layer's narrow wgrad alone.  Same launch
  // count, same workgroups, same arithmetic: the chain's last launch — 64-128 short workgroups, 6 us of launch floor —
  // is absorbed into a full wgrad launch (+~1 us), at the price of splitting one pair (8.3 + 7.9 instead of 14.5 us).
  const bool shifted = want_w && !input_grad && bwd_is_shifted(h, l, rows);
  if (qtrain != nullptr && !(shifted && lds_ok_of(l.L - 1))) return fail("internal: k_dgrad_qtrain needs the shifted schedule and an LDS-staged top layer");
  if (shifted) {
    {
      GemmBatch bd{}; bd.n = 1; bd.prob[0] = dgrad_of(l.L - 1);
      ScopedTiming t(h, 1, st);
      if (qtrain != nullptr) { bd.prob[0].Q = qtrain_seed; HIPCHK(dgrad_qtrain_launch(bd, *qtrain, st)); }
      else if (lds_ok_of(l.L - 1)) HIPCHK((dgrad_lds_launch<1, 1>(bd, st))); else HIPCHK((dgrad_direct_launch<1, 1>(bd, st)));
    }
    for (int i = l.L - 2; i >= 1; --i) {
      GemmBatch b{}; b.n = 2; b.prob[0] = dgrad_of(i); b.prob[1] = wgrad_of(i + 1);
      ScopedTiming t(h, 4, st);
      if (lds_ok_of(i)) HIPCHK((bwd_seq_launch<true>(b, st))); else HIPCHK((bwd_seq_launch<false>(b, st)));
      if (h->comm && h->dp_per_layer) RC(dp_reduce_slice(h, st, net, l.w_off[i + 1], layer_slice(i + 1)));
    }
    {
      GemmBatch b{}; b.n = 2; b.prob[0] = wgrad_of(1); b.prob[1] = wgrad_of(0);
      const HeadWgradRider none{};
      ScopedTiming t(h, 2, st);
      if (l.NH == 1) HIPCHK((wgrad_tail_launch<1>(b, rider ? *rider : none, st, tails))); else HIPCHK((wgrad_tail_launch<kNO>(b, rider ? *rider : none, st, tails)));
      if (h->comm && h->dp_per_layer) RC(dp_reduce_slice(h, st, net, l.w_off[0], layer_slice(0) + layer_slice(1)));
    }
    if (qrider) return fail("internal: the q-head rider found no carrier launch");
    return 0;
  }
  if (tails != nullptr) return fail("internal: the tails block rides in the shifted schedule's last launch only");
  for (int i = l.L - 1; i >= 0; --i) {
    GemmBatch bd{}, bw{};
    const bool need_dx = (i > 0 || input_grad);
    if (need_dx) bd.prob[bd.n++] = dgrad_of(i);
    if (want_w) bw.prob[bw.n++] = wgrad_of(i);
    const bool lds_ok = lds_ok_of(i);
    if (need_dx && want_w) {               // ONE workgroup type: its wgrad tile, then its dgrad tile (gemm_bwd_seq)
      GemmBatch b{}; b.n = 2; b.prob[0] = bd.prob[0]; b.prob[1] = bw.prob[0];
      ScopedTiming t(h, 4, st);
      // small minibatches / narrow layers: when the dgrad's 64x16 tiles and the wgrad's 64x64 tiles together still fit
      // the chip in one round, they run side by side on their own workgroups (one tile's chain per launch, not two)
      if (bwd_layer_is_pair(l, i, rows)) {
        if (lds_ok) HIPCHK((bwd_pair_direct_launch<1, true>(b, st))); else HIPCHK((bwd_pair_direct_launch<1, false>(b, st)));
      } else if (lds_ok) HIPCHK((bwd_seq_launch<true>(b, st)));
      else HIPCHK((bwd_seq_launch<false>(b, st)));
    } else if (need_dx && i == 0 && fuse != nullptr) {
      GemmProblem p = bd.prob[0];
      p.P += in_lo; p.C = nullptr; p.Pdim = 16; p.mask = nullptr;
      fuse->pr = p;
      const QHeadRider none{};
      ScopedTiming t(h, 1, st);
      HIPCHK(dqda_head_bwd_launch(*fuse, qrider ? *qrider : none, st));
      qrider = nullptr;
    } else if (need_dx && i == 0 && in_hi > in_lo && rows % 16 == 0) {
      GemmProblem& p = bd.prob[0];
      const int c0 = (in_lo / 16) * 16, c1 = std::min(l.kp[0], (in_hi + 15) / 16 * 16);
      p.P += c0; p.C += c0; p.Pdim = c1 - c0;
      if (p.mask) p.mask += c0;
      ScopedTiming t(h, 1, st);
      if (qrider) { HIPCHK(dgrad_narrow_qrider_launch(bd, *qrider, st)); qrider = nullptr; }
      else HIPCHK(dgrad_narrow_launch(bd, st));
    } else if (need_dx) {
      ScopedTiming t(h, 1, st);
      if (lds_ok) HIPCHK((dgrad_lds_launch<1, 1>(bd, st)));
      else HIPCHK((dgrad_direct_launch<1, 1>(bd, st)));
    } else {
      // wgrad alone = the first layer (K_in = 64 / 128 columns): 16-output tiles, 4x the workgroups
      ScopedTiming t(h, 2, st);
      if (rider) {                                    // + the head's dW / db as rider blocks (head_wgrad_can_ride)
        if (l.NH == 1) HIPCHK((wgrad_narrow_rider_launch<1>(bw, *rider, st))); else HIPCHK((wgrad_narrow_rider_launch<kNO>(bw, *rider, st)));
      } else HIPCHK((wgrad_narrow_launch<1>(bw, st)));
    }
    // data parallel, bucketed: layer i's dW/db are final once this launch has run -> start their
    // all-reduce on the communication stream while the chain continues with layer i-1
    if (want_w && h->comm && h->dp_per_layer) RC(dp_reduce_slice(h, st, net, l.w_off[i], layer_slice(i)));
  }
  if (qrider) return fail("internal: the q-head rider found no carrier launch");
  return 0;
}

// rows >= 1024: the bandwidth-tiled kernel pair; optionally emits the scaled fp16 panels itself
template <int NH>
int head_backward_big(H* h, hipStream_t st, HeadBwdArgs a, h16* dZ16, float scale16) {
  HeadBwdBigArgs b{}; b.a = a; b.dZ16 = dZ16; b.scale16 = scale16; b.slab2 = h->head_slab2;
  const int chunks = a.rows / 64;
  const int riders = a.q_out != nullptr ? chunks : 0;     // as many rider blocks again: 4 rows per block and round
  b.chunks = riders ? chunks : 0;
  const size_t lds = (size_t)(64 * NH + 4 * NH * 256) * sizeof(float);
  static bool prepared = false;
  if (!prepared) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_bwd_big<NH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); prepared = true; }
  hipLaunchKernelGGL((k_head_bwd_big<NH>), dim3(chunks + riders, a.H / 256), dim3(256), lds, st, b);
  HIPCHK(hipGetLastError());
  if (a.dW != nullptr) {
    hipLaunchKernelGGL((k_head_wred<NH>), dim3(a.H / 64, NH), dim3(256), 0, st, b, chunks);
    HIPCHK(hipGetLastError());
  }
  return 0;
}
inline bool head_big_ok(const H* h, int rows, int Hd) { return h->head_slab2 != nullptr && rows >= 1024 && rows % 64 == 0 && Hd % 256 == 0; }

template <int NH>
int head_backward(H* h, hipStream_t st, HeadBwdArgs a) {
  if (head_big_ok(h, a.rows, a.H)) return head_backward_big<NH>(h, st, a, nullptr, 1.0f);
  // row chunks: enough blocks to cover the chip a few times over, <= 64 rows per chunk
  const int RC = std::max(1, std::min(16, a.rows / 64));   // (64 chunks measured slower at B=4096: the last arriver's slab walk)
  const int rows_c = (a.rows + RC - 1) / RC;
  size_t lds = ((size_t)rows_c * NH + 16 * NH * 64 + 16) * sizeof(float);
  a.slab = h->head_slab; a.ticket = h->head_ticket;
  int ry = 0;                                               // extra grid rows for the q rider (16 rows per block)
  if (a.q_out != nullptr) { a.rc_blocks = RC; ry = ((a.rows + 15) / 16 + a.H / 64 - 1) / (a.H / 64); }
  hipLaunchKernelGGL((k_head_bwd<NH>), dim3(a.H / 64, RC + ry), dim3(1024), lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- the plan: which merged forms this learner's update takes ------------------------------------------------------------------
// ONE place decides (every predicate the launch sequence below branches on), run_phase / run_phase16 read it, and
// dqnhip_get_update_plan reports it together with the launch counts of a captured update: a predicate that silently stops matching at a
// BASELINE shape is a red test (tests/test_gpu_update_plan.py), not a slower bench.  A pure function of the learner's state (shapes,
// tuning flags, sharing, data-parallel mode): evaluated per call, never cached, so there is no stale copy to invalidate.
inline bool bwd16_has_carrier(const H* h, int net, int rows);      // (with tower_backward16, below)
UpdatePlan plan_of(const H* h) {
  const NetLayout &la = h->la, &lc = h->lc;
  const int B = h->B, L = h->L, Hh = la.dims[L], Hc = lc.dims[L];
  const int tf = h->cfg.tuning_flags;
  UpdatePlan p{};
  p.fp16 = h->fp16;
  p.dp = h->cfg.dp_world > 1 || h->dp_half || h->dp_shard;     // (a one-rank group with bf16 exchange / a sharded optimiser runs the N-rank code path)
  p.fused_seed = !(tf & DQNHIP_TUNE_SEPARATE_HEAD_SEED);
  if (h->fp16) {
    // fp16 learner (round 6): the head's dW / db are column-sum workgroups of the net's last backward launch (hgemm_group_db, HeadWsum)
    // when that launch exists; Step(1)'s k_head_q_train then also writes the scaled fp16 tower-top gradient — no head-backward launch
    // (below 1024 rows: beside the grouped wgrad's 200 one-per-CU tiles at 4096 rows the blocks cost more than the launches they replace)
    p.head_rides_c = bwd16_has_carrier(h, DQNHIP_CRITIC, B) && Hc % 64 == 0 && B < 1024;
    p.head_rides_a = bwd16_has_carrier(h, DQNHIP_ACTOR, B) && Hh % 64 == 0 && B < 1024;
    p.fuse_q = p.head_rides_c;
    p.tails_ride = p.dp && bwd16_has_carrier(h, DQNHIP_CRITIC, B) && bwd16_has_carrier(h, DQNHIP_ACTOR, B);
    // the critic's layer-0 dgrad (only its ten action columns are consumed), the inverting gradients and the actor heads' backward in
    // ONE launch (k_dqda_head_bwd<true>), as on the fp32 path; q(s, mu(s)) rides there
    p.fuse_head = p.head_rides_a && !(tf & DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD) && B % 16 == 0 && B < 1024 &&
                  h->S + 16 <= h->k16[1][0] && L >= 1 && lc.dims[1] % 64 == 0;
    return p;
  }
  p.shifted_c = bwd_is_shifted(h, lc, B); p.shifted_a = bwd_is_shifted(h, la, B);
  p.tails_ride = p.dp && p.shifted_c && p.shifted_a;        // data parallel: the tails block rides in each net's last backward launch
  // the head's own dW / db ride in the net's last backward launch (the first layer's narrow wgrad)
  p.head_rides_c = !head_big_ok(h, B, Hc) && head_wgrad_can_ride(lc, B);
  p.head_rides_a = !head_big_ok(h, B, Hh) && head_wgrad_can_ride(la, B);
  // Step(1)'s head arithmetic inside the critic's top-layer dgrad launch (k_dgrad_qtrain; one 16-column piece per lane: H / 16 <= 64)
  p.fuse_q = h->U3 != nullptr && !(tf & DQNHIP_TUNE_SEPARATE_Q_TRAIN) && p.head_rides_c && p.shifted_c && Hc >= 512 && Hc <= 1024 && Hc % 256 == 0;
  // dQ/da's last step, the inverting gradients and the actor heads' backward in ONE launch (k_dqda_head_bwd): 16 columns from the first
  // action column inside the panel row, fewer than 1024 rows (any tower-top width since round 6)
  p.fuse_head = p.head_rides_a && !(tf & DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD) && B % 16 == 0 && B < 1024 &&
                h->S + 16 <= lc.kp[0] && L >= 1 && lc.dims[1] % 64 == 0;
  // the first layer of critic(s, mu(s)) rides in the critic's optimiser launch (FirstLayerRider).  Data-parallel learners too (round 6):
  // the launch sits behind the critic's exchange point, its norm then comes from k_sumsq's partials; only a SHARDED optimiser — whose
  // pass covers 1/N of the arena — keeps the launch of its own
  p.critic_l0 = !h->dp_shard && h->shared_fl[DQNHIP_CRITIC] == 0 && !(tf & DQNHIP_TUNE_SEPARATE_FIRST_LAYER) &&
                (lc.kp[0] == 64 || lc.kp[0] == 128) && lc.dims[1] % 16 == 0 && B % 16 == 0 && B <= 512 && L >= 2 && lc.w_off[0] == 0 &&
                lc.b_off[0] == (size_t)lc.dims[1] * lc.kp[0];
  // Step(1)'s four first layers in one launch, critic_target's action half in the target actor's head kernel (first_layers_launch)
  p.first_layers_merged = h->Zs != nullptr && !(tf & DQNHIP_TUNE_SEPARATE_CRITIC_FIRST_LAYERS) && L >= 2 && B % 32 == 0 && B < 1024 && la.kp[0] < 512 &&
                          lc.kp[0] < 512 && la.dims[1] % 64 == 0 && lc.dims[1] % 64 == 0 && lc.dims[1] <= 1024 && round_up(h->S, 64) <= lc.kp[0];
  // inside a multi-update graph: the next update's gather rides in the critic's optimiser launch and its four first layers in the
  // actor's (k_adam_soft_fwd1_gather / k_adam_soft_l0).  Needs every piece those riders stand on.
  p.early_l0 = h->Xa_s2[1] != nullptr && !(tf & DQNHIP_TUNE_LATE_GATHER) && p.critic_l0 && p.first_layers_merged && h->shared_fl[DQNHIP_ACTOR] == 0 &&
               la.kp[0] == 64 && la.dims[1] % 16 == 0 && la.w_off[0] == 0 && la.b_off[0] == (size_t)la.dims[1] * la.kp[0];
  return p;
}
// ... while a multi-update graph is being captured (cap_u: the position of the update in it)
inline bool early_l0(const H* h) { return h->cap_u >= 0 && plan_of(h).early_l0; }

// The gather of one update (src/dqn.cpp:846-887).  pos: -1 outside multi-update graphs; else the update's position in the
// graph being captured (0: a launch of its own that also stores DevState::gbase; k >= 1: rides in update k-1's last launch)
GatherArgs gather_args(H* h, const int* idx_dev, int pos) {
  const NetLayout &la = h->la, &lc = h->lc;
  GatherArgs g{};
  g.ring = RO(h)->ring; g.rs = RO(h)->st; g.st = h->st; g.idx_in = idx_dev; g.seed = sample_key(h); g.B = h->B;
  if (h->fp16)
    // the gather writes the five minibatch panels in fp16 (what the GEMMs read: no conversion launch); the action columns
    // of the two critic panels the actor heads fill are zero here — the heads write mu / mu' straight into the panels
    g.o = GatherOut{nullptr, nullptr, la.kp[0], nullptr, nullptr, nullptr, lc.kp[0], h->mb_reward, h->mb_mc, h->mb_term, h->mb_idx,
                    h->act16[1][0], h->act16[0][0], h->act16[3][0], h->act16[4][0], h->act16[2][0]};
  else
    g.o = GatherOut{h->Xa_s2[pos > 0 && early_l0(h) ? (pos & 1) : 0], h->Xa_n, la.kp[0], h->Xc_tr, h->Xc_pl2[pos > 0 && early_l0(h) ? (pos & 1) : 0], h->Xc_nx, lc.kp[0],
                    h->mb_reward, h->mb_mc, h->mb_term, h->mb_idx};
  const int slot = pos > 0 ? (pos & 1) : 0;
  g.corr = h->st->adam_corr[slot]; g.soft_now = &h->st->soft_now[slot];
  g.beta1 = h->cfg.momentum; g.beta2 = h->cfg.momentum2; g.soft_update_freq = h->cfg.soft_update_freq;
  g.ahead = pos > 0 ? pos : -1; g.store_base = pos == 0 ? 1 : 0;
  if (h->chain_cap && pos > 0) { g.idx_in = h->idx_next_dev[pos & 1]; g.ahead = -2; }      // dqnhip_update_chained: the next update's indices, known one call ahead
  g.blocks = (h->B + 3) / 4 + 1;
  return g;
}

// clip + Adam + Net::Update + soft target update over arena floats [begin, end)
// corr_pre: the update's first launch (k_gather) has left this step's bias correction in DevState::adam_corr
int adam_launch(H* h, hipStream_t st, int net, const float* partial, int n_partial, size_t begin, size_t end, const TickArgs* tick,
                bool corr_pre, const FirstLayerRider* fl, const GatherArgs* early_gather, const NextL0* next_l0) {
  AdamArgs a{};
  const int slot = h->cap_u > 0 ? (h->cap_u & 1) : 0;      // DevState::adam_corr
  a.corr_pre = corr_pre ? &h->st->adam_corr[slot][net] : nullptr; a.soft_pre = corr_pre ? &h->st->soft_now[slot] : nullptr;
  a.w = h->w[net] + begin; a.g = h->g[net] + begin; a.m = h->m[net] + begin; a.v = h->v[net] + begin;
  a.wt = h->w[net + 2] + begin;
  if (h->fp16) { a.w16 = h->w16a[net] + begin; a.wt16 = h->w16a[net + 2] + begin; }
  const size_t sh = h->shared_fl[net];                 // shared prefix of this net's arena (floats)
  if (sh > begin) {
    a.w_sh = h->w_owner->w[net] + begin; a.wt_sh = h->w_owner->w[net + 2] + begin;
    a.n4_sh = (std::min(sh, end) - begin) / 4;
  }
  a.n4 = (end - begin) / 4; a.partial = partial; a.n_partial = n_partial;
  a.lr = net == DQNHIP_ACTOR ? h->cfg.actor_lr : h->cfg.critic_lr;
  a.beta1 = h->cfg.momentum; a.beta2 = h->cfg.momentum2; a.eps = h->cfg.delta;
  a.clip = h->cfg.clip_gradients; a.tau = (float)h->cfg.tau;
  a.soft_update_freq = h->cfg.soft_update_freq; a.which = net; a.st = h->st;
  if (tick) {
    if (!corr_pre) return fail("adam_launch: the update's bookkeeping needs the scalars k_gather leaves in DevState");
    a.tick_on = 1; a.tick = *tick;
  }
  ScopedTiming t(h, 3, st);
  LaunchTimer& lt = launch_timer();
  // 1536 blocks = 6 per CU, all resident at once (68 VGPRs: 7 waves per SIMD): with the loads hoisted above the prologue
  // same-box A/B gives 18.4 us per launch against 19.3 at 2048 (a second, short round of blocks), 19.4 at 1792, 18.7 at
  // 1280, 21.5 at 4096 (round 2, before the hoist: 512 .. 8192 within +-3 %, profiles/r02_adam_probe.txt)
  if (fl != nullptr || next_l0 != nullptr) {
    if (begin != 0 || (fl && tick != nullptr) || a.w_sh != nullptr || h->fp16 || !corr_pre) return fail("adam_launch: a first-layer rider needs the whole, unshared fp32 arena inside an update");
    a.skip4 = layout_of(h, net).w_off[1] / 4;
  }
  const int blocks = (int)std::min<size_t>((a.n4 - a.skip4 + 255) / 256 + (fl ? fl->blocks : 0), (size_t)1536);   // riders + the strided pass: what is resident at once
  if (next_l0 != nullptr) {
    // the actor's launch inside a multi-update graph that gathered early: the next update's four first layers ride here (k_adam_soft_l0)
    if (tick == nullptr) return fail("adam_launch: the next update's first layers ride in the update's last launch");
    const int riders = next_l0->a.blocks + next_l0->c.blocks + next_l0->ct.blocks;
    const int ablocks = std::min(blocks, std::max(1536 - riders, 1280));
    hipLaunchKernelGGL(k_adam_soft_l0, dim3(riders + ablocks), dim3(256), 0, st, a, next_l0->a, next_l0->c, next_l0->ct);
  }
  else if (fl != nullptr && early_gather != nullptr) {
    if (blocks <= fl->blocks) return fail("adam_launch: no optimiser workgroups beside the first-layer riders");
    const int ablocks = std::min(blocks, std::max(1536 - early_gather->blocks, 1280));
    if (fl->Kp == 64) hipLaunchKernelGGL(k_adam_soft_fwd1_gather<1>, dim3(early_gather->blocks + ablocks), dim3(256), 0, st, a, *fl, *early_gather);
    else hipLaunchKernelGGL(k_adam_soft_fwd1_gather<2>, dim3(early_gather->blocks + ablocks), dim3(256), 0, st, a, *fl, *early_gather);
  }
  else if (fl != nullptr) {
    if (blocks <= fl->blocks) return fail("adam_launch: no optimiser workgroups beside the first-layer riders");
    const bool timed = lt.start != nullptr;
    if (fl->Kp == 64) { if (timed) hipExtLaunchKernelGGL(k_adam_soft_fwd1<1>, dim3(blocks), dim3(256), 0, st, lt.start, lt.stop, 0, a, *fl); else hipLaunchKernelGGL(k_adam_soft_fwd1<1>, dim3(blocks), dim3(256), 0, st, a, *fl); }
    else { if (timed) hipExtLaunchKernelGGL(k_adam_soft_fwd1<2>, dim3(blocks), dim3(256), 0, st, lt.start, lt.stop, 0, a, *fl); else hipLaunchKernelGGL(k_adam_soft_fwd1<2>, dim3(blocks), dim3(256), 0, st, a, *fl); }
    if (timed) lt.start = lt.stop = nullptr;
  }
  else if (tick && h->cap_u >= 0 && h->cap_u + 1 < h->cap_n && !early_l0(h)) {
    // inside a multi-update graph: the next update's gather rides in this, the update's last launch (k_adam_soft_gather).
    // At small minibatches the grid stays at what is resident at once; a large minibatch's gather (1025 workgroups at 4096
    // rows) must not thin the optimiser's own grid — its blocks drain within a few us and the rest of the grid moves in
    // (511 optimiser blocks beside it: 35.4 against 27.3 us per launch at 4096 rows)
    const GatherArgs g = gather_args(h, nullptr, h->cap_u + 1);
    const int ablocks = std::min(blocks, std::max(1536 - g.blocks, 1280));
    hipLaunchKernelGGL(k_adam_soft_gather, dim3(g.blocks + ablocks), dim3(256), 0, st, a, g);
  }
  else if (lt.start) { hipExtLaunchKernelGGL(k_adam_soft, dim3(blocks), dim3(256), 0, st, lt.start, lt.stop, 0, a); lt.start = lt.stop = nullptr; }
  else hipLaunchKernelGGL(k_adam_soft, dim3(blocks), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// clip norm of the REDUCED gradient (data parallel); under DQNHIP_DP_HALF_GRADS the same pass widens the bf16
// transfer image back into the fp32 arena
int sumsq_launch(H* h, int net, size_t begin, size_t end) {
  const NetLayout& l = layout_of(h, net);
  if (end == 0) end = l.arena;
  if (h->dp_half) hipLaunchKernelGGL(k_sumsq_bf16, dim3(h->n_part_dp), dim3(256), 0, h->stream, (const uint16_t*)h->g16[net] + begin, h->g[net] + begin, (end - begin) / 4, h->part_dp);
  else hipLaunchKernelGGL(k_sumsq, dim3(h->n_part_dp), dim3(256), 0, h->stream, h->g[net] + begin, (end - begin) / 4, h->part_dp);
  HIPCHK(hipGetLastError());
  return 0;
}
// the optimiser step of one net inside a data-parallel update (phase 1: critic, phase 2: actor + bookkeeping); fl / early_gather /
// next_l0: the riders of adam_launch (replicated optimiser only: a sharded pass covers 1/N of the arena)
int dp_optimiser_step(H* h, hipStream_t st, int net, float* tail, const TickArgs* tick, const FirstLayerRider* fl = nullptr,
                      const GatherArgs* early_gather = nullptr, const NextL0* next_l0 = nullptr) {
  const NetLayout& l = layout_of(h, net);
  if (h->dp_shard) {
    if (fl || early_gather || next_l0) return fail("internal: riders in a sharded optimiser launch");
    // the exchange left this rank's slice of the reduced gradient in place and the group's sum of squares in tail[3]
    size_t lo, hi; shard_range(h, net, lo, hi);
    RC(adam_launch(h, st, net, tail + 3, 1, lo, hi, tick));
    return dp_allgather_weights(h, net);
  }
  RC(sumsq_launch(h, net));
  return adam_launch(h, st, net, h->part_dp, h->n_part_dp, 0, l.arena, tick, true, fl, early_gather, next_l0);
}

// ---- mixed-precision building blocks (hgemm.hip.h) --------------------------------------------

int hgemm_timed(H* h, hipStream_t st, const HGemm* gs, int n, int fam, int force = 0) {
  ScopedTiming t(h, fam, st);
  LaunchTimer& lt = launch_timer();
  hipEvent_t a = lt.start, b = lt.stop;
  lt.start = lt.stop = nullptr;
  HIPCHK(hgemm_launch_batch(gs, n, st, force, a, b));
  return 0;
}
int hgemm_timed(H* h, hipStream_t st, const HGemm& g, int fam) { return hgemm_timed(h, st, &g, 1, fam); }

// fp32 master weights of `net` -> fp16 mirror [N][kp].  The Adam pass keeps the mirrors current by itself; this
// runs after host-side weight changes (w16_dirty).
int sync_w16(H* h, hipStream_t st, int net) {
  const NetLayout& l = layout_of(h, net);
  const int kind = net & 1;
  Cvt16Batch b{};
  for (int i = 0; i < l.L; ++i) {
    cvt16_add(b, h->w[net] + l.w_off[i], l.kp[i], l.dims[i + 1], l.kp[i], h->w16[net][i], h->k16[kind][i], 1.0f);
    if (b.n == 8) { HIPCHK(cvt16_launch(b, st)); b = Cvt16Batch{}; }
  }
  HIPCHK(cvt16_launch(b, st));
  h->w16_dirty[net] = false;
  return 0;
}

HGemm fwd16_problem(H* h, int p, int net, int rows, int i) {
  const NetLayout& l = layout_of(h, net);
  const int kind = net & 1;
  HGemm g{};
  g.A = h->act16[p][i]; g.lda = h->k16[kind][i];
  g.B = h->w16[net][i]; g.ldb = h->k16[kind][i];
  g.M = rows; g.N = l.dims[i + 1]; g.K = h->k16[kind][i];
  g.C16 = h->act16[p][i + 1]; g.ldc16 = l.dims[i + 1];
  // (no fp32 copy of the tower top: the head kernels read the fp16 panel, as every tower layer reads its input)
  g.bias = h->w[net] + l.b_off[i]; g.relu = 1; g.scale32 = 1.0f;
  return g;
}
int tower_forward16(H* h, hipStream_t st, int p, int net, int rows) {
  const NetLayout& l = layout_of(h, net);
  for (int i = 0; i < l.L; ++i) RC(hgemm_timed(h, st, fwd16_problem(h, p, net, rows, i), 7));
  return 0;
}
// two independent passes of the same net kind, layer by layer in ONE launch each (the target and
// the online net: same shapes, different weights and inputs)
int tower_forward16_pair(H* h, hipStream_t st, int p0, int net0, int p1, int net1, int rows) {
  const NetLayout& l = layout_of(h, net0);
  for (int i = 0; i < l.L; ++i) {
    const HGemm gs[2] = {fwd16_problem(h, p0, net0, rows, i), fwd16_problem(h, p1, net1, rows, i)};
    RC(hgemm_timed(h, st, gs, 2, 7));
  }
  return 0;
}

// Tower backward in fp16 from dZ16[kind][L] (already scaled by `ls`).  want_w: dW (fp32, unscaled)
// into garena + bias gradients; input_grad: fp32 dZ32_0[rows][kp0] (unscaled).
// No transposed copy of any panel exists: the dgrad reads the weight mirror W[n][k_in] reduction-major (its rows ARE
// the reduction index), the wgrad reads dY[b][n_out] and X[b][k_in] reduction-major (hgemm.hip.h, HGemm::ta / tb).
// Schedule: the dgrad chain first (one launch per layer), then ALL wgrads of the net + the bias-gradient column
// sums in ONE launch (hgemm_group_db) — at that point every dZ panel is complete and the wgrads are independent.
// cfg.tuning_flags & DQNHIP_TUNE_FP16_WGRAD_PER_LAYER restores the per-layer form (a layer's dgrad + wgrad sharing a
// launch when both take the 64x64 tile; the bias sums riding in the first layer's wgrad launch).
// does a weights-wanted backward of this net end in a hgemm_group_db launch (the carrier of the head's dW / db riders and of a
// data-parallel learner's tails block)?  Grouped form: always; per-layer form: when the first layer's wgrad takes the 64 x 64 tile.
inline bool bwd16_has_carrier(const H* h, int net, int rows) {
  const NetLayout& l = layout_of(h, net);
  if (l.L <= kHGemmMax && rows >= kGroupMinRows && !(h->cfg.tuning_flags & DQNHIP_TUNE_FP16_WGRAD_PER_LAYER)) return true;
  HGemm g{}; g.ta = 1; g.tb = 1; g.M = l.dims[1]; g.N = h->k16[net & 1][0]; g.K = rows;
  return hgemm_uses_small_tile(g) && g.K % 128 == 0 && g.M % 64 == 0 && g.N % 64 == 0;
}
int tower_backward16(H* h, hipStream_t st, int net, int p, float* garena, float* dZ32_0, int rows,
                     bool want_w, bool input_grad, float ls, float* partial = nullptr, const HeadWsum* rider = nullptr,
                     const TailsArgs* tails = nullptr) {
  const NetLayout& l = layout_of(h, net);
  const int kind = net & 1;
  h16** dZ = h->dZ16[kind];
  const bool grouped = want_w && l.L <= kHGemmMax && rows >= kGroupMinRows && !(h->cfg.tuning_flags & DQNHIP_TUNE_FP16_WGRAD_PER_LAYER);
  HGemm gws[kMaxL];
  for (int i = l.L - 1; i >= 0; --i) {
    HGemm gd{};
    HGemm& gw = gws[i]; gw = HGemm{};
    const bool need_dx = i > 0 || input_grad;
    if (need_dx) {                         // dZ[i] = (dZ[i+1] . W_i) * lrelu'(act[i])
      HGemm& g = gd;
      g.A = dZ[i + 1]; g.lda = l.dims[i + 1];
      g.B = h->w16[net][i]; g.ldb = h->k16[kind][i]; g.tb = 1;
      g.M = rows; g.N = h->k16[kind][i]; g.K = l.dims[i + 1];
      if (i > 0) {
        // (the ReLU' operand is the whole fp16 activation panel although only its sign is used: a packed sign-bit form was
        // built and measured in round 4 — dgrad traffic 40 -> 33 MB per launch, duration unchanged, the forward's byte
        // stores +1 us per launch: profiles/r04_fp16_sign_mask.txt)
        g.mask = h->act16[p][i]; g.ldm = h->k16[kind][i];
        g.C16 = dZ[i]; g.ldc16 = h->k16[kind][i];
      } else {
        g.C32 = dZ32_0; g.ldc32 = l.kp[0]; g.n_valid32 = l.kp[0]; g.scale32 = 1.0f / ls;
      }
    }
    if (want_w) {                          // dW_i = dZ[i+1]^T . act[i]: both operands batch-major, dY[b][n_out], X[b][k_in]
      HGemm& g = gw;
      g.A = dZ[i + 1]; g.lda = l.dims[i + 1]; g.ta = 1;
      g.B = h->act16[p][i]; g.ldb = h->k16[kind][i]; g.tb = 1;
      g.M = l.dims[i + 1]; g.N = h->k16[kind][i]; g.K = rows;
      g.C32 = garena + l.w_off[i]; g.ldc32 = l.kp[i]; g.n_valid32 = l.kp[i]; g.scale32 = 1.0f / ls;
      if (partial) g.sumsq_partial = partial + l.part_off[i];      // clip-norm share of this layer's dW (unscaled)
    }
    if (grouped) { if (need_dx) RC(hgemm_timed(h, st, gd, 8)); continue; }
    // per-layer form: both read dZ[i+1] and neither reads the other's output — at small minibatches (both on the
    // 64x64 split-K tile) they share one launch
    if (need_dx && want_w && hgemm_uses_small_tile(gd) && hgemm_uses_small_tile(gw) && gd.K % 128 == 0 && gw.K % 128 == 0) {
      const HGemm gs[2] = {gd, gw};
      RC(hgemm_timed(h, st, gs, 2, 8, 2));      // both on the 64x64 tile, as each would be alone
    } else {
      if (need_dx) RC(hgemm_timed(h, st, gd, 8));
      if (want_w && (i > 0 || need_dx)) RC(hgemm_timed(h, st, gw, 9));
    }
  }
  if (!want_w) return 0;
  // db_i = column sums of dZ[i+1] [rows][n_out], one workgroup per 64 columns
  Db16Batch db{}; db.scale = 1.0f / ls; db.sumsq_partial = partial ? partial + l.part_db : nullptr;
  int db_blocks = 0;
  for (int i = 0; i < l.L; ++i) { db.d[db.n++] = Db16{dZ[i + 1], l.dims[i + 1], l.dims[i + 1], rows, garena + l.b_off[i], db_blocks}; db_blocks += l.dims[i + 1] / 64; }
  ScopedTiming t(h, 9, st);
  LaunchTimer& lt = launch_timer();
  hipEvent_t e0 = lt.start, e1 = lt.stop;
  lt.start = lt.stop = nullptr;
  if (grouped) {
    // 128x128 tiles: a quarter of the operand bytes per FLOP of the 64x64 split-K tile (fp16 mode guarantees
    // hidden % 128 == 0, minibatch % 128 == 0 and a 128-wide first panel, so every wgrad tiles)
    HIPCHK(hgemm_group_db_launch(gws, l.L, true, db, db_blocks, st, e0, e1, rider, tails));
  } else if (!input_grad && hgemm_uses_small_tile(gws[0]) && gws[0].K % 128 == 0) {
    // per-layer form: the first layer's wgrad (few tiles, long reduction) carries the column sums
    HIPCHK(hgemm_group_db_launch(gws, 1, false, db, db_blocks, st, e0, e1, rider, tails));
  } else {
    if (rider != nullptr || tails != nullptr) return fail("internal: the fp16 backward found no carrier launch for its riders");
    if (!input_grad) HIPCHK(hgemm_launch(gws[0], st, 0, e0, e1));
    hipLaunchKernelGGL(k_db16_cols<0>, dim3(db_blocks), dim3(256), 0, st, db);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int run_phase16(H* h, int phase, const int* idx_dev) {
  const int B = h->B, L = h->L;
  const NetLayout &la = h->la, &lc = h->lc;
  const UpdatePlan P = plan_of(h);
  const bool dp = P.dp;
  const float inv_batch = 1.0f / (float)(B * h->cfg.dp_world);
  float* actor_tail = (h->dp_half || h->dp_shard) ? h->dp_tails + 4 : h->g[0] + la.arena;
  float* critic_tail = (h->dp_half || h->dp_shard) ? h->dp_tails : h->g[1] + lc.arena;
  const int Hh = la.dims[L], Hc = lc.dims[L];
  hipStream_t st = h->stream;
  const bool split = phase == 10;
  // single learner: the clip norm comes from the partial sums the wgrad / bias-gradient / head workgroups leave
  // behind (as on the fp32 path); data-parallel ranks need the norm of the REDUCED gradient: k_sumsq
  const bool part16 = !dp;
  if (phase == 11) {
    HeadArgs hA{}; hA.X16 = h->act16[1][L]; hA.ldx = Hh; hA.H = Hh; hA.rows = B;
    hA.W = wat(h, DQNHIP_ACTOR, la.hw_off); hA.b = wat(h, DQNHIP_ACTOR, la.hb_off);
    hA.out16 = h->aout16; hA.xc = nullptr; hA.ldxc = lc.kp[0]; hA.xc_col = h->S;
    hA.xc16 = h->act16[4][0]; hA.ldxc16 = h->k16[1][0];
    RC(tower_forward16(h, st, 1, DQNHIP_ACTOR, B));
    RC((head_forward<kNO, HEAD_ACTOR>(h, st, hA)));
    return 0;
  }
  if (phase == 0 || phase == 10) {
    if (h->cap_u <= 0) {       // (later updates of a multi-update graph: the gather rode in the previous update's last launch)
      const GatherArgs g = gather_args(h, idx_dev, h->cap_u);
      hipLaunchKernelGGL(k_gather, dim3(g.blocks), dim3(256), 0, st, g);
      HIPCHK(hipGetLastError());
    }
    if (split) RC(tower_forward16(h, st, 0, DQNHIP_ACTOR_TARGET, B));
    else RC(tower_forward16_pair(h, st, 0, DQNHIP_ACTOR_TARGET, 1, DQNHIP_ACTOR, B));
    HeadArgs hAT{}; hAT.X16 = h->act16[0][L]; hAT.ldx = Hh; hAT.H = Hh; hAT.rows = B;
    hAT.W = wat(h, DQNHIP_ACTOR_TARGET, la.hw_off); hAT.b = wat(h, DQNHIP_ACTOR_TARGET, la.hb_off);
    hAT.out16 = h->aout_t16; hAT.xc = nullptr; hAT.ldxc = lc.kp[0]; hAT.xc_col = h->S;     // (fp32 panels: unused in fp16 mode)
    HeadArgs hA{}; hA.X16 = h->act16[1][L]; hA.ldx = Hh; hA.H = Hh; hA.rows = B;
    hA.W = wat(h, DQNHIP_ACTOR, la.hw_off); hA.b = wat(h, DQNHIP_ACTOR, la.hb_off);
    hA.out16 = h->aout16; hA.xc = nullptr; hA.ldxc = lc.kp[0]; hA.xc_col = h->S;
    hAT.xc16 = h->act16[2][0]; hAT.ldxc16 = h->k16[1][0]; hA.xc16 = h->act16[4][0]; hA.ldxc16 = h->k16[1][0];
    if (split) RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT)));
    else RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT, &hA)));
    RC(tower_forward16_pair(h, st, 2, DQNHIP_CRITIC_TARGET, 3, DQNHIP_CRITIC, B));
    {
      HeadTrainArgs t{};
      t.Xt16 = h->act16[2][L]; t.Wt = wat(h, DQNHIP_CRITIC_TARGET, lc.hw_off); t.bt = wat(h, DQNHIP_CRITIC_TARGET, lc.hb_off);
      t.X16 = h->act16[3][L]; t.W = wat(h, DQNHIP_CRITIC, lc.hw_off); t.b = wat(h, DQNHIP_CRITIC, lc.hb_off);
      t.H = Hc; t.rows = B; t.reward = h->mb_reward; t.mc = h->mb_mc; t.term = h->mb_term;
      t.q_target = h->q_t; t.q = h->q1; t.y = h->y; t.dq = h->dq; t.loss_partial = h->loss_partial;
      t.gamma = h->cfg.gamma; t.beta = h->cfg.beta; t.inv_batch = inv_batch; t.st = h->st;
      // round 6: with the head's dW / db riding in the net's last backward launch, the scaled fp16 tower-top gradient comes out of
      // this launch too (HeadTrainArgs::dZ16) — Step(1) has no head-backward launch (as on the fp32 path since round 3)
      if (P.fuse_q) { t.dZ16 = h->dZ16[1][L]; t.scale16 = h->ls_c; }
      hipLaunchKernelGGL(k_head_q_train, dim3((B + 3) / 4), dim3(256), 0, st, t);
      HIPCHK(hipGetLastError());
    }
    const TailsArgs tails_c{(const float*)h->loss_partial, h->n_head_blocks, (const double*)nullptr, 0, inv_batch, critic_tail, (float*)nullptr, &h->st->flags, 0};
    {
      HeadBwdArgs a{}; a.dyh = h->dq; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X416 = h->act16[3][L];
      a.H = Hc; a.rows = B; a.dZ = nullptr; a.dW = h->g[1] + lc.hw_off; a.db = h->g[1] + lc.hb_off;
      a.partial = h->part[1] + lc.part_off[L];
      const HeadWsum r{h->dq, 1, h->act16[3][L], Hc, B, a.dW, a.db, a.partial, 1, Hc / 64};
      if (P.head_rides_c) {}                                       // (dZ16 came out of k_head_q_train, dW / db come from the riders)
      else if (head_big_ok(h, B, Hc)) { a.dZ = nullptr; RC(head_backward_big<1>(h, st, a, h->dZ16[1][L], h->ls_c)); }
      else { a.dZ16 = h->dZ16[1][L]; a.scale16 = h->ls_c; RC(head_backward<1>(h, st, a)); }
      RC(tower_backward16(h, st, DQNHIP_CRITIC, 3, h->g[1], nullptr, B, true, false, h->ls_c, part16 ? h->part[1] : nullptr, P.head_rides_c ? &r : nullptr,
                          P.tails_ride ? &tails_c : nullptr));
    }
    if (dp && !P.tails_ride) {
      hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, tails_c);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  if (phase == 1) {
    // the Adam pass writes the fp16 mirrors of the critic and its target itself
    if (part16) RC(adam_launch(h, st, 1, h->part[1], lc.n_part, 0, lc.arena));
    else RC(dp_optimiser_step(h, st, 1, critic_tail, nullptr));
    // As on the fp32 path: the seed of the dq = -1 pass comes out of the top layer's forward epilogue (HGemm::seed_w, the
    // scaled fp16 panel the dgrad chain reads) and q(s, mu(s)) rides in a later launch-floor launch — here the actor heads'
    // backward (HeadBwdArgs::qr_*).  DQNHIP_TUNE_SEPARATE_HEAD_SEED: the head-backward launch of their own.
    const bool fused_seed = P.fused_seed;
    for (int i = 0; i < L; ++i) {
      HGemm g = fwd16_problem(h, 4, DQNHIP_CRITIC, B, i);
      if (fused_seed && i == L - 1) { g.seed_w = wat(h, DQNHIP_CRITIC, lc.hw_off); g.CS16 = h->dZ16[1][L]; g.ldcs16 = Hc; g.seed_scale = h->ls_q; }
      RC(hgemm_timed(h, st, g, 7));
    }
    if (!fused_seed) {
      // q(s, mu(s)) rides in the dq = -1 head launch (rider blocks)
      HeadBwdArgs a{}; a.dyh = nullptr; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X416 = h->act16[4][L];
      a.H = Hc; a.rows = B; a.dZ = nullptr;
      a.q_bias = wat(h, DQNHIP_CRITIC, lc.hb_off); a.q_out = h->q2; a.qsum_partial = h->q_partial;
      if (head_big_ok(h, B, Hc)) { a.dZ = nullptr; RC(head_backward_big<1>(h, st, a, h->dZ16[1][L], h->ls_q)); }
      else { a.dZ16 = h->dZ16[1][L]; a.scale16 = h->ls_q; RC(head_backward<1>(h, st, a)); }
    }
    RC(tower_backward16(h, st, DQNHIP_CRITIC, 4, nullptr, h->dZc[0], B, false, !P.fuse_head, h->ls_q));       // (fuse_head: layers L-1 .. 1 only)
    if (P.fuse_head) {
      DqdaHeadArgs fz{};
      fz.aout16 = h->aout16; fz.dA16 = h->dA16; fz.W = wat(h, DQNHIP_ACTOR, la.hw_off); fz.H = Hh; fz.rows = B;
      fz.t16 = NarrowTile16{h->w16[DQNHIP_CRITIC][0] + h->S, h->k16[1][0], h->dZ16[1][1], lc.dims[1], lc.dims[1]}; fz.inv_ls = 1.0f / h->ls_q;
      fz.X416 = h->act16[1][L]; fz.dZ16 = h->dZ16[0][L]; fz.scale16 = h->ls_a;
      // (DQNHIP_TUNE_SEPARATE_HEAD_SEED: q(s, mu(s)) came out of the dq = -1 head launch — no rider blocks)
      const QHeadRider qr{nullptr, wat(h, DQNHIP_CRITIC, lc.hw_off), wat(h, DQNHIP_CRITIC, lc.hb_off), h->q2, h->q_partial, Hc, B, fused_seed ? (B + 3) / 4 : 0, h->act16[4][L]};
      HIPCHK(dqda_head_bwd_launch(fz, qr, st));
    }
    {
      HeadBwdArgs a{}; a.dXc = h->dZc[0]; a.ldx = lc.kp[0]; a.S = h->S; a.aout16 = h->aout16; a.dA16 = h->dA16;
      a.W = wat(h, DQNHIP_ACTOR, la.hw_off); a.X416 = h->act16[1][L]; a.H = Hh; a.rows = B; a.dZ = nullptr;
      if (fused_seed) {
        a.q_bias = wat(h, DQNHIP_CRITIC, lc.hb_off); a.q_out = h->q2; a.qsum_partial = h->q_partial;
        a.qr_W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.qr_X416 = h->act16[4][L]; a.qr_H = Hc;
      }
      a.dW = h->g[0] + la.hw_off; a.db = h->g[0] + la.hb_off; a.partial = h->part[0] + la.part_off[L];
      const HeadWsum r{h->dA16, kAP, h->act16[1][L], Hh, B, a.dW, a.db, a.partial, kNO, Hh / 64};   // dA16: the post-invert diffs this launch leaves
      if (P.head_rides_a) { a.dW = nullptr; a.db = nullptr; a.partial = nullptr; }
      if (P.fuse_head) {}                                        // (k_dqda_head_bwd<true> above did all of it)
      else if (head_big_ok(h, B, Hh)) { a.dZ = nullptr; RC(head_backward_big<kNO>(h, st, a, h->dZ16[0][L], h->ls_a)); }
      else { a.dZ16 = h->dZ16[0][L]; a.scale16 = h->ls_a; RC(head_backward<kNO>(h, st, a)); }
      const TailsArgs tails_a{(const float*)nullptr, 0, (const double*)h->q_partial, B, inv_batch, (float*)nullptr, actor_tail, &h->st->flags, 0};
      RC(tower_backward16(h, st, DQNHIP_ACTOR, 1, h->g[0], nullptr, B, true, false, h->ls_a, part16 ? h->part[0] : nullptr, P.head_rides_a ? &r : nullptr,
                          P.tails_ride ? &tails_a : nullptr));
      if (dp && !P.tails_ride) {
        hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, tails_a);
        HIPCHK(hipGetLastError());
      }
    }
    return 0;
  }
  if (phase == 2) {
    const TickArgs tick{h->st, critic_tail, actor_tail, (const float*)h->loss_partial, h->n_head_blocks,
                        dp ? (const double*)nullptr : (const double*)h->q_partial, B, (float)(B * h->cfg.dp_world), h->stats_dev};
    if (part16) RC(adam_launch(h, st, 0, h->part[0], la.n_part, 0, la.arena, &tick));
    else RC(dp_optimiser_step(h, st, 0, actor_tail, &tick));        // + iteration counters / statistics
    h->h_actor_iter += 1; h->h_critic_iter += 1;
    return 0;
  }
  return fail("phase must be 0, 1 or 2 (got %d)", phase);
}

int sync_dirty16(H* h) {
  if (!h->fp16) return 0;
  for (int net = 0; net < 4; ++net) if (h->w16_dirty[net]) RC(sync_w16(h, h->stream, net));
  return 0;
}

// ---- the update, in three phases (see dqnhip.h) ---------------------------------
// Step(1)'s four first tower layers in ONE launch (round 5): actor_target(s'), actor(s), critic(s, a) — each exactly what
// layer_forward(…, 0) computes — and the STATE half of critic_target(s', mu'(s'))'s first layer (K = the state columns; no bias, no
// ReLU; into h->Zs).  Its action half is a rank-10 update per row that the target actor's head kernel applies itself
// (HeadArgs::l1_*), so the launch of the critics' first layers between the heads and the critics' second layer is gone.
// with_actor false (the data-parallel overlap form): the online actor's whole forward runs later (phase 11) — three problems.
int first_layers_launch(H* h, hipStream_t st, int rows, bool with_actor) {
  const NetLayout &la = h->la, &lc = h->lc;
  GemmBatch b{}; b.n = with_actor ? 4 : 3;
  auto fill = [&](GemmProblem& p, int net, const NetLayout& l, const float* X, float* Y, int kred, bool finish) {
    p.P = wat(h, net, l.w_off[0]); p.ldp = l.kp[0];
    p.Q = X; p.ldq = l.kp[0];
    p.C = Y; p.ldc = l.kp[1];
    p.Pdim = l.dims[1]; p.Qdim = rows; p.Kred = kred;
    p.bias = finish ? wat(h, net, l.b_off[0]) : nullptr; p.relu = finish ? 1 : 0;
  };
  fill(b.prob[0], DQNHIP_ACTOR_TARGET, la, h->act[0][0], h->act[0][1], la.kp[0], true);
  fill(b.prob[1], DQNHIP_CRITIC, lc, h->act[3][0], h->act[3][1], lc.kp[0], true);
  fill(b.prob[2], DQNHIP_CRITIC_TARGET, lc, h->act[2][0], h->Zs, round_up(h->S, 64), false);
  b.prob[2].xcopy_dst = h->Wact_t; b.prob[2].xcopy_col = h->S; b.prob[2].xcopy_n = kNO;
  if (with_actor) fill(b.prob[3], DQNHIP_ACTOR, la, h->act[1][0], h->act[1][1], la.kp[0], true);
  ScopedTiming t(h, 6, st);
  HIPCHK((fwd_direct_launch<4, 2>(b, st)));
  return 0;
}
int run_phase(H* h, int phase, const int* idx_dev) {
  const UpdatePlan P = plan_of(h);
  select_panels(h, early_l0(h) ? (h->cap_u & 1) : 0);
  if (h->fp16) return run_phase16(h, phase, idx_dev);
  const int B = h->B, L = h->L;
  const NetLayout &la = h->la, &lc = h->lc;
  const bool dp = P.dp;
  const float inv_batch = 1.0f / (float)(B * h->cfg.dp_world);
  float* actor_tail = (h->dp_half || h->dp_shard) ? h->dp_tails + 4 : h->g[0] + la.arena;
  float* critic_tail = (h->dp_half || h->dp_shard) ? h->dp_tails : h->g[1] + lc.arena;
  const int Hh = la.dims[L], Hc = lc.dims[L];
  hipStream_t st = h->stream;
  const bool split = phase == 10;          // phase 10 = phase 0 without the online actor's forward, 11 = that forward
  if (phase == 11) {
    FwdPass pA{DQNHIP_ACTOR, &la, h->act[1]};
    HeadArgs hA{}; hA.X = h->act[1][L]; hA.ldx = Hh; hA.H = Hh; hA.rows = B;
    hA.W = wat(h, DQNHIP_ACTOR, la.hw_off); hA.b = wat(h, DQNHIP_ACTOR, la.hb_off);
    hA.out16 = h->aout16; hA.xc = h->Xc_pl; hA.ldxc = lc.kp[0]; hA.xc_col = h->S;
    RC(tower_forward(h, st, &pA, 1, B));
    RC((head_forward<kNO, HEAD_ACTOR>(h, st, hA)));
    return 0;
  }
  if (phase == 0 || phase == 10) {
    // 1-2: sample + gather (src/dqn.cpp:846-887)
    if (h->cap_u <= 0) {       // (later updates of a multi-update graph: the gather rode in the previous update's last launch)
      const GatherArgs g = gather_args(h, idx_dev, h->cap_u);
      hipLaunchKernelGGL(k_gather, dim3(g.blocks), dim3(256), 0, st, g);
      HIPCHK(hipGetLastError());
    }
    FwdPass pAT{DQNHIP_ACTOR_TARGET, &la, h->act[0]}, pA{DQNHIP_ACTOR, &la, h->act[1]};
    FwdPass pCT{DQNHIP_CRITIC_TARGET, &lc, h->act[2]}, pC1{DQNHIP_CRITIC, &lc, h->act[3]};
    HeadArgs hAT{}; hAT.X = h->act[0][L]; hAT.ldx = Hh; hAT.H = Hh; hAT.rows = B;
    hAT.W = wat(h, DQNHIP_ACTOR_TARGET, la.hw_off); hAT.b = wat(h, DQNHIP_ACTOR_TARGET, la.hb_off);
    hAT.out16 = h->aout_t16; hAT.xc = h->Xc_nx; hAT.ldxc = lc.kp[0]; hAT.xc_col = h->S;
    HeadArgs hA{}; hA.X = h->act[1][L]; hA.ldx = Hh; hA.H = Hh; hA.rows = B;
    hA.W = wat(h, DQNHIP_ACTOR, la.hw_off); hA.b = wat(h, DQNHIP_ACTOR, la.hb_off);
    hA.out16 = h->aout16; hA.xc = h->Xc_pl; hA.ldxc = lc.kp[0]; hA.xc_col = h->S;
    // Step(1)'s head arithmetic (q', q, TD target, loss, dq, dZ_L) inside the critic's top-layer dgrad launch (k_dgrad_qtrain)
    // instead of a launch of its own: the online critic's top forward layer then also leaves U = (-w_h) lrelu'(x_L)
    const bool fuse_q = P.fuse_q;
    if (fuse_q) {
      pC1.seed_w = wat(h, DQNHIP_CRITIC, lc.hw_off); pC1.seed_out = h->U3;
      pC1.dot_w = pC1.seed_w; pC1.dot_out = h->qdot[1];
      pCT.dot_w = wat(h, DQNHIP_CRITIC_TARGET, lc.hw_off); pCT.dot_out = h->qdot[0];
    }
    FwdPass cp[2] = {pCT, pC1};
    // all four first layers of Step(1) in the update's first GEMM launch, critic_target's action half in the head kernel (first_layers_launch);
    // DQNHIP_TUNE_SEPARATE_CRITIC_FIRST_LAYERS: the critics' first layers in a launch of their own behind the heads
    const bool merged_l0 = P.first_layers_merged;
    if (merged_l0) {
      if (!(early_l0(h) && h->cap_u > 0)) RC(first_layers_launch(h, st, B, !split));     // (else: they rode in the previous update's last launch, k_adam_soft_l0)
      hAT.l1_zs = h->Zs; hAT.l1_wt = h->Wact_t;
      hAT.l1_b = wat(h, DQNHIP_CRITIC_TARGET, lc.b_off[0]); hAT.l1_y = h->act[2][1]; hAT.l1_ld = lc.kp[1]; hAT.l1_n = lc.dims[1];
    }
    if (split) {
      // data-parallel overlap form: the online actor's forward (phase 11) is left out so that it can
      // run while the critic gradients are being all-reduced
      RC(tower_forward(h, st, &pAT, 1, B, merged_l0 ? 1 : 0));
      RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT)));
    } else {
      // actor_target(s') [src/dqn.cpp:889-891] and actor(s) [:910-911, pre-update weights] layer by layer in one
      // launch each, then critic_target(s', mu'(s')) and the critic(s, a) train forward [:904] likewise
      FwdPass ap[2] = {pAT, pA};
      RC(tower_forward(h, st, ap, 2, B, merged_l0 ? 1 : 0));
      RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT, &hA)));     // both actors' heads in one launch
    }
    RC(tower_forward(h, st, cp, 2, B, merged_l0 ? 1 : 0));
    HeadTrainArgs qt_args{};
    {
      HeadTrainArgs t{};
      t.Xt = h->act[2][L]; t.Wt = wat(h, DQNHIP_CRITIC_TARGET, lc.hw_off); t.bt = wat(h, DQNHIP_CRITIC_TARGET, lc.hb_off);
      t.X = h->act[3][L]; t.W = wat(h, DQNHIP_CRITIC, lc.hw_off); t.b = wat(h, DQNHIP_CRITIC, lc.hb_off);
      t.H = Hc; t.rows = B; t.reward = h->mb_reward; t.mc = h->mb_mc; t.term = h->mb_term;
      t.q_target = h->q_t; t.q = h->q1; t.y = h->y; t.dq = h->dq; t.loss_partial = h->loss_partial;
      t.gamma = h->cfg.gamma; t.beta = h->cfg.beta; t.inv_batch = inv_batch; t.st = h->st;
      // with the head's dW / db riding in the net's last backward launch, the head's dZ comes out of this launch too
      if (P.head_rides_c) t.dZ = h->dZc[L];
      t.pdt = h->qdot[0]; t.pd = h->qdot[1];
      qt_args = t;
      if (!fuse_q) {
        hipLaunchKernelGGL(k_head_q_train, dim3((B + 3) / 4), dim3(256), 0, st, t);
        HIPCHK(hipGetLastError());
      }
    }
    // critic backward (rest of Step(1)): head (dgrad + ReLU' + wgrad fused), then tower; wgrad
    // writes (beta=0) so ClearParamDiffs/ZeroGradParameters (src/dqn.cpp:63-78, 908-909) vanish
    {
      HeadBwdArgs a{}; a.dyh = h->dq; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X4 = h->act[3][L];
      a.H = Hc; a.rows = B; a.dZ = h->dZc[L]; a.dW = h->g[1] + lc.hw_off; a.db = h->g[1] + lc.hb_off;
      a.partial = h->part[1] + lc.part_off[L];
      // the head's own gradients ride in the net's last backward launch (the first layer's narrow wgrad)
      const bool ride = P.head_rides_c;
      HeadWgradRider r{h->dq, 1, h->act[3][L], Hc, B, a.dW, a.db, a.partial, Hc / kRiderCW};
      if (!ride) RC(head_backward<1>(h, st, a));          // (riding: dZ came out of k_head_q_train, dW / db come from the rider)
      // data parallel: [loss, q, flag] tails for the exchange — one more block of the backward's last launch, or a launch of its own
      const TailsArgs tails_c{(const float*)h->loss_partial, h->n_head_blocks, (const double*)nullptr, 0, inv_batch, critic_tail, (float*)nullptr, &h->st->flags, 0};
      RC(tower_backward(h, st, lc, DQNHIP_CRITIC, h->g[1], h->part[1], h->act[3], h->dZc, B, true, false, 0, -1, ride ? &r : nullptr, nullptr, nullptr,
                        fuse_q ? &qt_args : nullptr, fuse_q ? h->U3 : nullptr, P.tails_ride ? &tails_c : nullptr));
      if (dp && !P.tails_ride) {
        hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, tails_c);
        HIPCHK(hipGetLastError());
      }
    }
    return 0;
  }
  if (phase == 1) {
    // ClipGradients + Adam + Net::Update of the critic, soft update of critic_target fused (one pass)
    FwdPass pC2{DQNHIP_CRITIC, &lc, h->act[4]};
    h->act[4][0] = h->Xc_pl;
    // the first layer of critic(s, mu(s)) rides in the critic's optimiser launch (FirstLayerRider: the workgroups that own W1 run it
    // on the weights they have just stepped); DQNHIP_TUNE_SEPARATE_FIRST_LAYER: a launch of its own (same bits)
    const bool ride_l0 = P.critic_l0;
    if (ride_l0) {
      const FirstLayerRider fl{h->Xc_pl, lc.kp[0], h->act[4][1], lc.kp[1], B, lc.kp[0], lc.dims[1], lc.dims[1] / 16};
      // inside a multi-update graph the NEXT update's gather rides here too (its panels: the other parity), so that its first layers
      // can ride in the actor's launch
      const bool eg = early_l0(h) && h->cap_u + 1 < h->cap_n;
      const GatherArgs g = eg ? gather_args(h, nullptr, h->cap_u + 1) : GatherArgs{};
      if (dp) RC(dp_optimiser_step(h, st, 1, critic_tail, nullptr, &fl, eg ? &g : nullptr));
      else RC(adam_launch(h, st, 1, h->part[1], lc.n_part, 0, lc.arena, nullptr, true, &fl, eg ? &g : nullptr));
    }
    else if (dp) RC(dp_optimiser_step(h, st, 1, critic_tail, nullptr));
    else RC(adam_launch(h, st, 1, h->part[1], lc.n_part, 0, lc.arena));
    // The seed of BackwardFrom(q_values_layer) [:918-923] — q diff = -1 per row, taken through the head and the top
    // layer's ReLU, input gradient only (the reference's discarded critic dW, SURVEY a11, is never computed) — does not
    // depend on q: it comes out of the top tower layer's forward epilogue, and q(s, mu(s)) itself [:913-916], which only
    // the statistics read, rides in the chain's last launch.  DQNHIP_TUNE_SEPARATE_HEAD_SEED: the head-backward launch
    // that used to sit between the forward and the backward chain (same arithmetic, one launch more).
    const bool fused_seed = P.fused_seed;
    if (fused_seed) { pC2.seed_w = wat(h, DQNHIP_CRITIC, lc.hw_off); pC2.seed_out = h->dZc[L]; }
    RC(tower_forward(h, st, &pC2, 1, B, ride_l0 ? 1 : 0));   // critic(s, mu(s)), UPDATED weights [:913-916]
    const QHeadRider qr{h->act[4][L], wat(h, DQNHIP_CRITIC, lc.hw_off), wat(h, DQNHIP_CRITIC, lc.hb_off), h->q2, h->q_partial, Hc, B, (B + 3) / 4};
    if (!fused_seed) {
      HeadBwdArgs a{}; a.dyh = nullptr; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X4 = h->act[4][L];
      a.H = Hc; a.rows = B; a.dZ = h->dZc[L];
      a.q_bias = wat(h, DQNHIP_CRITIC, lc.hb_off); a.q_out = h->q2; a.qsum_partial = h->q_partial;
      RC(head_backward<1>(h, st, a));
    }
    // dQ/da's last step, the inverting gradients (src/dqn.cpp:924-957) and the actor heads' backward (src/dqn.cpp:960-963) share
    // ONE launch (k_dqda_head_bwd) when the actor head's own gradients ride in the actor's last backward launch and the shapes
    // allow it (16 columns from the first action column inside the panel row, H a multiple of 256, fewer than 1024 rows);
    // DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD: the narrow dgrad launch + k_head_bwd<10> (same arithmetic, one launch more)
    const bool ride_a = P.head_rides_a;
    const bool fuse_head = P.fuse_head;
    DqdaHeadArgs fz{};
    fz.aout16 = h->aout16; fz.dA16 = h->dA16; fz.W = wat(h, DQNHIP_ACTOR, la.hw_off); fz.X4 = h->act[1][L]; fz.dZ = h->dZa[L]; fz.H = Hh; fz.rows = B;
    RC(tower_backward(h, st, lc, DQNHIP_CRITIC, nullptr, nullptr, h->act[4], h->dZc, B, false, true, h->S, h->S + kNO, nullptr, fused_seed ? &qr : nullptr,
                      fuse_head ? &fz : nullptr));
    // inverting gradients (src/dqn.cpp:924-957) + actor heads backward (src/dqn.cpp:960-963)
    {
      HeadBwdArgs a{}; a.dXc = h->dZc[0]; a.ldx = lc.kp[0]; a.S = h->S; a.aout16 = h->aout16; a.dA16 = h->dA16;
      a.W = wat(h, DQNHIP_ACTOR, la.hw_off); a.X4 = h->act[1][L]; a.H = Hh; a.rows = B; a.dZ = h->dZa[L];
      a.dW = h->g[0] + la.hw_off; a.db = h->g[0] + la.hb_off; a.partial = h->part[0] + la.part_off[L];
      const bool ride = ride_a;
      HeadWgradRider r{h->dA16, kAP, h->act[1][L], Hh, B, a.dW, a.db, a.partial, Hh / kRiderCW};   // dA16: the post-invert diffs this launch leaves
      if (ride) { a.dW = nullptr; a.db = nullptr; a.partial = nullptr; }
      if (!fuse_head) RC(head_backward<kNO>(h, st, a));
      const TailsArgs tails_a{(const float*)nullptr, 0, (const double*)h->q_partial, B, inv_batch, (float*)nullptr, actor_tail, &h->st->flags, 0};
      RC(tower_backward(h, st, la, DQNHIP_ACTOR, h->g[0], h->part[0], h->act[1], h->dZa, B, true, false, 0, -1, ride ? &r : nullptr, nullptr, nullptr, nullptr, nullptr,
                        P.tails_ride ? &tails_a : nullptr));
      if (dp && !P.tails_ride) {
        hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, tails_a);
        HIPCHK(hipGetLastError());
      }
    }
    return 0;
  }
  if (phase == 2) {
    // the actor's optimiser pass is the update's last launch: its block 0 also publishes
    // (critic_loss, avg_q) and advances the iteration / sampling counters
    const TickArgs tick{h->st, critic_tail, actor_tail, (const float*)h->loss_partial, h->n_head_blocks,
                        dp ? (const double*)nullptr : (const double*)h->q_partial, B, (float)(B * h->cfg.dp_world), h->stats_dev};
    if (early_l0(h) && h->cap_u + 1 < h->cap_n) {
      // the next update's first layers ride here: its panels (gathered in this update's critic launch) are those of the other parity
      const int pn = (h->cap_u + 1) & 1;
      NextL0 n{};
      n.a = ActorL0{h->Xa_s2[pn], h->Xa_n, la.kp[0], h->act[1][1], h->act[0][1], la.kp[1], B, la.dims[1], la.dims[1] / 16};
      n.c = PlainL0{h->w[DQNHIP_CRITIC] + lc.w_off[0], lc.kp[0], h->w[DQNHIP_CRITIC] + lc.b_off[0], h->Xc_tr, lc.kp[0], h->act[3][1], lc.kp[1],
                    B, lc.kp[0], lc.dims[1], nullptr, 0, 0, lc.dims[1] / 16};
      n.ct = PlainL0{h->w[DQNHIP_CRITIC_TARGET] + lc.w_off[0], lc.kp[0], nullptr, h->Xc_nx, lc.kp[0], h->Zs, lc.kp[1],
                     B, round_up(h->S, 64), lc.dims[1], h->Wact_t, h->S, kNO, lc.dims[1] / 16};
      if (dp) RC(dp_optimiser_step(h, st, 0, actor_tail, &tick, nullptr, nullptr, &n));
      else RC(adam_launch(h, st, 0, h->part[0], la.n_part, 0, la.arena, &tick, true, nullptr, nullptr, &n));
    }
    else if (dp) RC(dp_optimiser_step(h, st, 0, actor_tail, &tick));
    else RC(adam_launch(h, st, 0, h->part[0], la.n_part, 0, la.arena, &tick));
    h->h_actor_iter += 1; h->h_critic_iter += 1;
    return 0;
  }
  return fail("phase must be 0, 1, 2, 10 or 11 (got %d)", phase);
}

// re-read (head,size) after the env front-end appended episodes on the device
int refresh_ring(H* h) {
  if (!RO(h)->ring_stale) return 0;
  int hs[2];
  HIPCHK(hipMemcpyAsync(hs, RO(h)->st, sizeof hs, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  RO(h)->h_head = hs[0]; RO(h)->h_size = hs[1]; RO(h)->ring_stale = false;
  return 0;
}

int stage_indices(H* h, const int32_t* idx_host, const int** idx_dev) {
  *idx_dev = nullptr;
  if (idx_host || RO(h)->h_size < 1) RC(refresh_ring(h));
  if (RO(h)->h_size < 1) return fail("replay memory is empty");
  if (idx_host) {
    for (int i = 0; i < h->B; ++i)
      if (idx_host[i] < 0 || idx_host[i] >= RO(h)->h_size)
        return fail("sampled index %d = %d out of range [0,%lld)", i, idx_host[i], RO(h)->h_size);
    // the pinned staging buffer may still be in flight from the previous update
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(h->idx_pinned, idx_host, h->B * sizeof(int));
    *idx_dev = h->idx_pinned_dev;           // the gather reads them from the pinned host buffer itself
  }
  return 0;
}

int ensure_stage(H* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return 0;
  if (h->stage_dev) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->stage_dev)); h->stage_dev = nullptr; }
  bytes = round_up_z(bytes, 1 << 20);
  HIPCHK(hipMalloc(&h->stage_dev, bytes));
  h->stage_bytes = bytes;
  return 0;
}

// acting-time activation scratch for `rows` rows of the widest net
int ensure_act(H* h, int rows) {
  size_t need = 0;
  for (int i = 0; i <= h->L; ++i) need += (size_t)rows * std::max(h->la.kp[i], h->lc.kp[i]);
  need += (size_t)rows * (kAP + 1);
  if (need <= h->act_floats) return 0;
  if (h->act_buf) { HIPCHK(hipStreamSynchronize(h->stream)); HIPCHK(hipFree(h->act_buf)); h->act_buf = nullptr; }
  HIPCHK(hipMalloc(&h->act_buf, need * sizeof(float)));
  h->act_floats = need;
  return 0;
}


void drop_graphs(H* h) {
  for (auto& g : h->graph_exec) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  for (auto& g : h->graph_small) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  if (h->dp_graph) { hipGraphExecDestroy(h->dp_graph); h->dp_graph = nullptr; }
  if (h->dp_graph_n) { hipGraphExecDestroy(h->dp_graph_n); h->dp_graph_n = nullptr; }
  h->dp_graph_failed = false; h->dp_graph_n_failed = false; h->graph_failed = false;
}
int to_bf16_launch(H* h, int net) {
  hipLaunchKernelGGL(k_to_bf16, dim3(1024), dim3(256), 0, h->stream, (const float*)h->g[net], layout_of(h, net).arena / 4, h->g16[net]);
  HIPCHK(hipGetLastError());
  return 0;
}
int shard_scal_launch(H* h, float* tail) {
  hipLaunchKernelGGL(k_shard_scal, dim3(1), dim3(256), 0, h->stream, (const float*)h->part_dp, h->n_part_dp, tail);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace dqnhip_host

// ================================ C ABI =========================================
extern "C" {

void dqnhip_default_config(dqnhip_config* c, int32_t state_size) {
  memset(c, 0, sizeof *c);
  c->struct_size = (int32_t)sizeof *c;
  c->minibatch = 32;                       // src/dqn.hpp:19
  c->state_size = state_size;
  c->num_hidden = 4;                       // src/dqn.cpp:425,449
  c->hidden[0] = 1024; c->hidden[1] = 512; c->hidden[2] = 256; c->hidden[3] = 128;
  c->replay_capacity = 500000;             // src/dqn.cpp:25
  c->soft_update_freq = 1;                 // :23
  c->gamma = .99; c->beta = .5; c->tau = .001;   // :24, :31, :22
  c->actor_lr = 0.00001f; c->critic_lr = 0.001f; // src/dqn_main.cpp:33-34
  c->momentum = .95f; c->momentum2 = .999f;      // src/dqn_main.cpp:31-32
  c->delta = 1e-8f;                        // Caffe SolverParameter.delta default
  c->clip_gradients = 10.f;                // src/dqn_main.cpp:35
  c->device = 0; c->dp_world = 1; c->dp_rank = 0; c->use_graph = 0; c->seed = 1;
}

const char* dqnhip_last_error(void) { return g_err.c_str(); }
// used by snapshot.cpp (same library, different translation unit) to report through the same channel
int dqnhip_internal_set_error(const char* msg) { g_err = msg ? msg : ""; return 1; }

int dqnhip_get_config(dqnhip_handle h, dqnhip_config* out) {
  if (!h || !out) return fail("null argument");
  *out = h->cfg;
  out->stream = nullptr; out->grad_arena = nullptr; out->grad_arena_bytes = 0;
  return 0;
}

size_t dqnhip_grad_arena_bytes(const dqnhip_config* cfg) {
  if (validate(cfg)) return 0;
  NetLayout la, lc;
  layout_init(la, cfg->state_size, *cfg, true);
  layout_init(lc, cfg->state_size + kNO, *cfg, false);
  return grad_arena_floats(la, lc) * sizeof(float);
}

static int create_impl(H* h, const dqnhip_config* cfg);

int dqnhip_create(const dqnhip_config* cfg, dqnhip_handle* out) {
  if (!out) return fail("out is null");
  *out = nullptr;
  RC(validate(cfg));
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d not available (%d visible)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  H* h = new H();
  const int rc = create_impl(h, cfg);
  if (rc) {                                  // free whatever was allocated; keep the first error message
    const std::string msg = g_err;
    dqnhip_destroy(h);
    g_err = msg;
    return rc;
  }
  *out = h;
  return 0;
}

static int create_impl(H* h, const dqnhip_config* cfg) {
  h->cfg = *cfg; h->B = cfg->minibatch; h->S = cfg->state_size; h->L = cfg->num_hidden;
  layout_init(h->la, h->S, *cfg, true);
  layout_init(h->lc, h->S + kNO, *cfg, false);
  if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
  else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
  const int B = h->B, L = h->L;
  auto dalloc = [&](float** p, size_t n) -> int {
    HIPCHK(hipMalloc(p, n * sizeof(float)));
    HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(float), h->stream));
    return 0;
  };
  for (int i = 0; i < 4; ++i) RC(dalloc(&h->w[i], layout_of(h, i).arena));
  for (int i = 0; i < 2; ++i) { RC(dalloc(&h->m[i], layout_of(h, i).arena)); RC(dalloc(&h->v[i], layout_of(h, i).arena)); }
  const size_t gfl = grad_arena_floats(h->la, h->lc);
  if (cfg->grad_arena) {
    if (cfg->grad_arena_bytes < gfl * sizeof(float)) return fail("grad_arena too small: %zu < %zu", cfg->grad_arena_bytes, gfl * sizeof(float));
    h->grad_base = (float*)cfg->grad_arena;
    HIPCHK(hipMemsetAsync(h->grad_base, 0, gfl * sizeof(float), h->stream));
  } else { RC(dalloc(&h->grad_base, gfl)); h->own_grad = true; }
  h->g[0] = h->grad_base; h->g[1] = h->grad_base + h->la.arena + 64;
  // replay ring
  Ring& r = h->ring;
  r.cap = cfg->replay_capacity; r.S = h->S; r.SP = round_up(h->S, 64);
  RC(dalloc(&r.state, (size_t)r.cap * r.SP)); RC(dalloc(&r.next, (size_t)r.cap * r.SP));
  RC(dalloc(&r.act, (size_t)r.cap * kAP)); RC(dalloc(&r.reward, r.cap)); RC(dalloc(&r.mc, r.cap));
  HIPCHK(hipMalloc(&r.term, r.cap)); HIPCHK(hipMemsetAsync(r.term, 0, r.cap, h->stream));
  HIPCHK(hipMalloc(&h->st, sizeof(DevState))); HIPCHK(hipMemsetAsync(h->st, 0, sizeof(DevState), h->stream));
  HIPCHK(hipMalloc(&h->done_counter, sizeof(int))); HIPCHK(hipMemsetAsync(h->done_counter, 0, sizeof(int), h->stream));
  // panels and activations
  RC(dalloc(&h->Xa_s, (size_t)B * h->la.kp[0])); RC(dalloc(&h->Xa_n, (size_t)B * h->la.kp[0]));
  RC(dalloc(&h->Xc_tr, (size_t)B * h->lc.kp[0])); RC(dalloc(&h->Xc_pl, (size_t)B * h->lc.kp[0]));
  h->Xa_s2[0] = h->Xa_s; h->Xc_pl2[0] = h->Xc_pl;
  if (!h->fp16) { RC(dalloc(&h->Xa_s2[1], (size_t)B * h->la.kp[0])); RC(dalloc(&h->Xc_pl2[1], (size_t)B * h->lc.kp[0])); }
  else { h->Xa_s2[1] = nullptr; h->Xc_pl2[1] = nullptr; }
  RC(dalloc(&h->Xc_nx, (size_t)B * h->lc.kp[0]));
  h->act[0][0] = h->Xa_n; h->act[1][0] = h->Xa_s; h->act[2][0] = h->Xc_nx; h->act[3][0] = h->Xc_tr; h->act[4][0] = h->Xc_pl;
  for (int p = 0; p < 5; ++p)
    for (int i = 1; i <= L; ++i) RC(dalloc(&h->act[p][i], (size_t)B * layout_of(h, p >= 2).kp[i]));
  for (int i = 0; i <= L; ++i) { RC(dalloc(&h->dZa[i], (size_t)B * h->la.kp[i])); RC(dalloc(&h->dZc[i], (size_t)B * h->lc.kp[i])); }
  RC(dalloc(&h->U3, (size_t)B * h->lc.kp[L]));        // the training pass's head-seed panel (k_dgrad_qtrain)
  for (int j = 0; j < 2; ++j) RC(dalloc(&h->qdot[j], (size_t)B * (h->lc.kp[L] / 16)));
  RC(dalloc(&h->Wact_t, (size_t)kNO * h->lc.dims[1]));  // critic_target's first-layer action-column weights, transposed (GemmProblem::xcopy_dst)
  RC(dalloc(&h->Zs, (size_t)B * h->lc.kp[1]));   // the state half of critic_target's first layer (first_layers_launch)
  RC(dalloc(&h->mb_reward, B)); RC(dalloc(&h->mb_mc, B)); RC(dalloc(&h->mb_term, B));
  HIPCHK(hipMalloc(&h->mb_idx, B * sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&h->idx_pinned, B * sizeof(int), hipHostMallocMapped));
  HIPCHK(hipHostMalloc((void**)&h->pinned_stats, 64, hipHostMallocMapped));
  memset(h->pinned_stats, 0, 64);
  { void* d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, h->idx_pinned, 0)); h->idx_pinned_dev = (const int*)d;
    HIPCHK(hipHostGetDevicePointer(&d, h->pinned_stats, 0)); h->stats_dev = (float*)d; }
  RC(dalloc(&h->aout_t16, (size_t)B * kAP)); RC(dalloc(&h->aout16, (size_t)B * kAP)); RC(dalloc(&h->dA16, (size_t)B * kAP));
  RC(dalloc(&h->q_t, B)); RC(dalloc(&h->q1, B)); RC(dalloc(&h->q2, B)); RC(dalloc(&h->y, B)); RC(dalloc(&h->dq, B));
  h->n_head_blocks = (B + 3) / 4;
  RC(dalloc(&h->loss_partial, h->n_head_blocks));
  HIPCHK(hipMalloc(&h->q_partial, B * sizeof(double)));
  HIPCHK(hipMemsetAsync(h->q_partial, 0, B * sizeof(double), h->stream));
  RC(dalloc(&h->part[0], h->la.n_part)); RC(dalloc(&h->part[1], h->lc.n_part));
  h->n_part_dp = 1024; RC(dalloc(&h->part_dp, h->n_part_dp));
  {
    const int Hmax = std::max(h->la.dims[L], h->lc.dims[L]);
    RC(dalloc(&h->head_slab, (size_t)64 * (Hmax / 64) * kNO * 64 + 64 * 16));
    if (B >= 1024 && B % 64 == 0) RC(dalloc(&h->head_slab2, (size_t)(B / 64) * kNO * Hmax + (size_t)(B / 64) * 16));
    HIPCHK(hipMalloc(&h->head_ticket, (Hmax / 64) * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->head_ticket, 0, (Hmax / 64) * sizeof(int), h->stream));
  }
  if (cfg->precision == DQNHIP_FP16) {
    h->fp16 = true;
    const float user = cfg->loss_scale > 0.f ? cfg->loss_scale : 1.0f;
    h->ls_c = 16.0f * (float)(B * cfg->dp_world) * user;   // dq = (q-y)/B_global: back to O(q-y)
    h->ls_q = 4096.0f * user;
    h->ls_a = 16384.0f * user;
    auto halloc = [&](h16** p, size_t n) -> int {
      HIPCHK(hipMalloc(p, n * sizeof(h16)));
      HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(h16), h->stream));
      h->allocs16.push_back((void*)*p);
      return 0;
    };
    for (int kind = 0; kind < 2; ++kind) {
      const NetLayout& l = kind ? h->lc : h->la;
      for (int i = 0; i <= L; ++i) h->k16[kind][i] = l.kp[i];
    }
    for (int net = 0; net < 4; ++net) {
      const NetLayout& l = layout_of(h, net);
      RC(halloc(&h->w16a[net], l.arena));
      for (int i = 0; i < L; ++i) h->w16[net][i] = h->w16a[net] + l.w_off[i];
    }
    for (int p = 0; p < 5; ++p) {
      const int kind = p >= 2;
      for (int i = 0; i <= L; ++i) RC(halloc(&h->act16[p][i], (size_t)B * h->k16[kind][i]));
    }
    for (int kind = 0; kind < 2; ++kind)
      for (int i = 0; i <= L; ++i) RC(halloc(&h->dZ16[kind][i], (size_t)B * h->k16[kind][i]));
    HIPCHK(hgemm_prepare_all());
  }
  // weights: gaussian(std 0.01), zero bias (src/dqn.cpp:350-352); targets = hard copy (:660-661)
  {
    std::mt19937_64 rng(cfg->seed * 0x9E3779B97F4A7C15ull + 12345);
    std::normal_distribution<float> nd(0.0f, 0.01f);
    for (int net = 0; net < 2; ++net) {
      const NetLayout& l = layout_of(h, net);
      std::vector<float> dense(l.dense, 0.0f), arena;
      size_t d = 0;
      for (int i = 0; i < l.L; ++i) {
        const size_t nw = (size_t)l.dims[i + 1] * l.dims[i];
        for (size_t e = 0; e < nw; ++e) dense[d + e] = nd(rng);
        d += nw + l.dims[i + 1];
      }
      const int Hh = l.dims[l.L];
      if (net == 0) {
        for (size_t e = 0; e < (size_t)kNA * Hh; ++e) dense[d + e] = nd(rng);
        d += (size_t)kNA * Hh + kNA;
        for (size_t e = 0; e < (size_t)kNP * Hh; ++e) dense[d + e] = nd(rng);
      } else {
        for (size_t e = 0; e < (size_t)Hh; ++e) dense[d + e] = nd(rng);
      }
      dense_to_arena(l, dense.data(), arena);
      HIPCHK(hipMemcpyAsync(h->w[net], arena.data(), l.arena * sizeof(float), hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      HIPCHK(hipMemcpyAsync(h->w[net + 2], h->w[net], l.arena * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
  }
  HIPCHK(direct_prepare(gemm_bwd_seq<true>, 4 * 16 * 64 * 16 + 4 * 16 * 16));
  HIPCHK(direct_prepare(gemm_bwd_seq<false>, 4 * 16 * 64 * 16 + 4 * 16 * 16));
  HIPCHK(direct_prepare(gemm_wgrad_tail<1>, 80 * 1024));
  HIPCHK(direct_prepare(gemm_wgrad_tail<kNO>, 80 * 1024));
  HIPCHK(direct_prepare((gemm_bwd_pair_direct<1, true>), 4 * 16 * 64 * 16 + 4 * 16 * 16));
  HIPCHK(direct_prepare((gemm_bwd_pair_direct<1, false>), 4 * 16 * 64 * 16 + 4 * 16 * 16));
  HIPCHK(direct_prepare(gemm_fwd_lds<4, 2, false>, 4 * 2 * 6 * 512 * 4));
  HIPCHK(direct_prepare(gemm_fwd_lds<4, 2, true>, 4 * 2 * 6 * 512 * 4));
  RC(sync_dirty16(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int dqnhip_destroy(dqnhip_handle h) {
  if (!h) return 0;
  if (h->sharers > 0) return fail("dqnhip_destroy: %d learner(s) still share this learner's layers / replay memory; destroy them first", h->sharers);
  if (h->w_owner) h->w_owner->sharers -= 1;
  if (h->ring_owner) h->ring_owner->sharers -= 1;
  if (h->ring_ev) hipEventDestroy(h->ring_ev);
  hipSetDevice(h->cfg.device);
  dp_destroy_impl(h, false);
  hipStreamSynchronize(h->stream);
  for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (auto& g : h->graph_exec) if (g) hipGraphExecDestroy(g);
  for (auto& g : h->graph_small) if (g) hipGraphExecDestroy(g);
  for (int i = 0; i < 2; ++i) {
    if (h->pipe_ev[i]) hipEventDestroy(h->pipe_ev[i]);
    if (h->pipe_idx_pinned[i]) hipHostFree(h->pipe_idx_pinned[i]);
    if (h->pipe_stats[i]) hipHostFree(h->pipe_stats[i]);
  }
  for (int i = 0; i < 4; ++i) hipFree(h->w[i]);
  for (int i = 0; i < 2; ++i) { hipFree(h->m[i]); hipFree(h->v[i]); hipFree(h->part[i]); }
  if (h->own_grad) hipFree(h->grad_base);
  hipFree(h->ring.state); hipFree(h->ring.next); hipFree(h->ring.act); hipFree(h->ring.reward);
  hipFree(h->ring.mc); hipFree(h->ring.term); hipFree(h->st); hipFree(h->done_counter);
  hipFree(h->Xa_s2[0]); hipFree(h->Xa_s2[1]); hipFree(h->Xa_n); hipFree(h->Xc_tr); hipFree(h->Xc_pl2[0]); hipFree(h->Xc_pl2[1]); hipFree(h->Xc_nx);
  for (int p = 0; p < 5; ++p) for (int i = 1; i <= h->L; ++i) hipFree(h->act[p][i]);
  for (int i = 0; i <= h->L; ++i) { hipFree(h->dZa[i]); hipFree(h->dZc[i]); }
  hipFree(h->mb_reward); hipFree(h->mb_mc); hipFree(h->mb_term); hipFree(h->mb_idx); hipFree(h->U3); hipFree(h->qdot[0]); hipFree(h->qdot[1]); hipFree(h->Zs); hipFree(h->Wact_t);
  hipHostFree(h->idx_pinned); hipHostFree(h->pinned_stats);
  for (int i = 0; i < 2; ++i) if (h->idx_next_pinned[i]) hipHostFree(h->idx_next_pinned[i]);
  hipFree(h->aout_t16); hipFree(h->aout16); hipFree(h->dA16);
  hipFree(h->q_t); hipFree(h->q1); hipFree(h->q2); hipFree(h->y); hipFree(h->dq);
  hipFree(h->loss_partial); hipFree(h->q_partial); hipFree(h->part_dp); hipFree(h->head_slab); hipFree(h->head_ticket); if (h->head_slab2) hipFree(h->head_slab2);
  for (void* p : h->allocs16) hipFree(p);
  if (h->stage_dev) hipFree(h->stage_dev);
  if (h->shard_total) hipFree(h->shard_total);
  if (h->act_buf) hipFree(h->act_buf);
  if (h->own_stream) hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

// ---- update -----------------------------------------------------------------------

// kMultiU updates per replay of graph_exec[4].  Two consecutive hipGraphLaunch calls leave the GPU idle for ~8.4 us between the
// last kernel of one and the first kernel of the next (kernel trace, profiles/r04_graph_gap.txt; two instances of the
// graph launched alternately: the same) - 2.8 % of a 300-us update; inside a graph the same boundary is a plain kernel boundary.
// Inside it the gather of update u + 1 rides in update u's last launch (adam_launch, DevState::gbase).
static int capture_graph(H* h, int which, const int* idx_fixed = nullptr, int n_multi = kMultiU, hipGraphExec_t* out = nullptr) {
  // Capture phases 0,1,2 once; replays re-read every changing scalar from DevState
  // and (which == 1) the indices from the fixed pinned buffer through a memcpy node; which == 2, 3: the indices
  // are already in the given device buffer (dqnhip_update_pipelined's two slots).
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  const int* idx_dev = idx_fixed;
  if (which == 1 || which == 5) idx_dev = h->idx_pinned_dev;
  const int it_a = h->h_actor_iter, it_c = h->h_critic_iter;
  h->cap_n = which == 4 ? n_multi : kMultiU;       // (chain graphs: always a successor's riders)
  for (int u = 0; u < (which == 4 ? n_multi : 1); ++u) {
    // which >= 5 (dqnhip_update_chained): ONE update captured as position 0 (head: own gather), 1 or 2 (continued at parity 1 / 0) of
    // a multi-update graph, its riders reading the next update's explicit indices
    h->cap_u = which == 4 ? u : which >= 5 ? which - 5 : -1;
    h->chain_cap = which >= 5;
    for (int p = 0; p < 3 && !rc; ++p) rc = run_phase(h, p, idx_dev);
  }
  h->cap_u = -1; h->chain_cap = false; h->cap_n = kMultiU;
  select_panels(h, 0);
  h->h_actor_iter = it_a; h->h_critic_iter = it_c;   // capture does not execute
  hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(out != nullptr ? out : &h->graph_exec[which], graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) return fail("hipGraphInstantiate: %s", hipGetErrorString(e));
  return 0;
}

// kernel nodes of a captured sequence
static int count_kernel_nodes(hipGraph_t g, int* out) {
  size_t cnt = 0;
  HIPCHK(hipGraphGetNodes(g, nullptr, &cnt));
  std::vector<hipGraphNode_t> nodes(cnt);
  if (cnt) HIPCHK(hipGraphGetNodes(g, nodes.data(), &cnt));
  int k = 0;
  for (hipGraphNode_t nd : nodes) {
    hipGraphNodeType t;
    HIPCHK(hipGraphNodeGetType(nd, &t));
    if (t == hipGraphNodeTypeKernel) ++k;
  }
  *out = k;
  return 0;
}
// the kernels `updates` consecutive updates launch — as a multi-update graph captures them (multi) or stand-alone.  A capture that
// is thrown away: nothing executes, nothing is instantiated.
static int count_launches(H* h, bool multi, int updates, int* out) {
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  const int it_a = h->h_actor_iter, it_c = h->h_critic_iter;
  int rc = 0;
  for (int u = 0; u < updates && !rc; ++u) {
    h->cap_u = multi ? u : -1;
    for (int p = 0; p < 3 && !rc; ++p) rc = run_phase(h, p, nullptr);
  }
  h->cap_u = -1;
  select_panels(h, 0);
  h->h_actor_iter = it_a; h->h_critic_iter = it_c;
  const std::string msg = g_err;
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); g_err = msg; return rc; }
  if (e != hipSuccess) return fail("hipStreamEndCapture (plan): %s", hipGetErrorString(e));
  rc = count_kernel_nodes(graph, out);
  hipGraphDestroy(graph);
  return rc;
}

int dqnhip_get_update_plan(dqnhip_handle h, dqnhip_update_plan* out) {
  if (!h || !out) return fail("null argument");
  if (out->struct_size != (int32_t)sizeof(dqnhip_update_plan)) return fail("dqnhip_update_plan.struct_size %d != %zu (ABI mismatch)", out->struct_size, sizeof(dqnhip_update_plan));
  if (h->next_phase != 0) return fail("dqnhip_get_update_plan: a phased update is in progress (next phase %d)", h->next_phase);
  if (h->timing) return fail("dqnhip_get_update_plan: kernel timing is on (timed launches are not captured)");
  HIPCHK(hipSetDevice(h->cfg.device));
  const UpdatePlan p = plan_of(h);
  memset(out, 0, sizeof *out);
  out->struct_size = (int32_t)sizeof *out;
  out->forms = (p.fp16 ? DQNHIP_PLAN_FP16 : 0) | (p.dp ? DQNHIP_PLAN_DATA_PARALLEL : 0) | (p.shifted_c ? DQNHIP_PLAN_BWD_SHIFTED_CRITIC : 0) |
               (p.shifted_a ? DQNHIP_PLAN_BWD_SHIFTED_ACTOR : 0) | (p.head_rides_c ? DQNHIP_PLAN_HEAD_WGRAD_RIDES_CRITIC : 0) |
               (p.head_rides_a ? DQNHIP_PLAN_HEAD_WGRAD_RIDES_ACTOR : 0) | (p.fuse_q ? DQNHIP_PLAN_Q_TRAIN_IN_DGRAD : 0) |
               (p.fused_seed ? DQNHIP_PLAN_HEAD_SEED_FUSED : 0) | (p.fuse_head ? DQNHIP_PLAN_DQDA_HEAD_BWD : 0) |
               (p.critic_l0 ? DQNHIP_PLAN_CRITIC_L0_RIDES : 0) | (p.first_layers_merged ? DQNHIP_PLAN_FIRST_LAYERS_MERGED : 0) |
               (p.early_l0 ? DQNHIP_PLAN_EARLY_GATHER_L0 : 0) | (p.tails_ride ? DQNHIP_PLAN_DP_TAILS_RIDE : 0);
  out->updates_per_graph = kMultiU;
  // per net and update: one all-reduce (per-layer buckets: one per tower layer + the head slice; bf16 exchange: the tails travel in a
  // second call beside the actor's; sharded optimiser: reduce-scatter + the 4-float all-reduce + 2 (fp16 learner: 4) all-gathers)
  if (h->comm) out->collectives = h->dp_shard ? 2 * (2 + (h->fp16 ? 4 : 2)) : h->dp_per_layer ? 2 * (h->L + 1) : h->dp_half ? 3 : 2;
  if (h->dp_shard) return 0;
  int n1 = 0, nf = 0, n2 = 0;
  RC(count_launches(h, false, 1, &n1));
  RC(count_launches(h, true, 1, &nf));
  RC(count_launches(h, true, 2, &n2));
  out->launches_single = n1; out->launches_graph_first = nf; out->launches_in_graph = n2 - nf;
  return 0;
}

int dqnhip_update_async(dqnhip_handle h, const int32_t* idx_host) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->cfg.dp_world > 1) return fail("dqnhip_update_async: dp_world > 1 requires dqnhip_update_phase + all-reduce (or dqnhip_dp_update)");
  if (h->dp_half) return fail("dqnhip_update_async: this learner exchanges bf16 gradients (DQNHIP_DP_HALF_GRADS): use dqnhip_dp_update");
  if (h->dp_shard) return fail("dqnhip_update_async: this learner's optimiser is sharded over its group (DQNHIP_DP_SHARD_OPT): use dqnhip_dp_update");
  if (h->next_phase != 0) return fail("dqnhip_update_async: a phased update is in progress (next phase %d)", h->next_phase);
  RingUse ring_use(h);
  RC(sync_dirty16(h));
  if (h->cfg.use_graph && !h->timing && !h->graph_failed) {
    if (idx_host || RO(h)->h_size < 1) RC(refresh_ring(h));
    if (RO(h)->h_size < 1) return fail("replay memory is empty");
    const int which = idx_host ? 1 : 0;
    if (idx_host) {
      for (int i = 0; i < h->B; ++i)
        if (idx_host[i] < 0 || idx_host[i] >= RO(h)->h_size) return fail("sampled index out of range");
      HIPCHK(hipStreamSynchronize(h->stream));
      memcpy(h->idx_pinned, idx_host, h->B * sizeof(int));
    }
    if (!h->graph_exec[which]) {
      if (capture_graph(h, which)) { h->graph_failed = true; }
    }
    if (h->graph_exec[which]) {
      HIPCHK(hipGraphLaunch(h->graph_exec[which], h->stream));
      h->h_actor_iter += 1; h->h_critic_iter += 1;
      return 0;
    }
  }
  const int* idx_dev = nullptr;
  RC(stage_indices(h, idx_host, &idx_dev));
  for (int p = 0; p < 3; ++p) RC(run_phase(h, p, idx_dev));
  return 0;
}

int dqnhip_update_async_n(dqnhip_handle h, int32_t n) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  if (n < 0) return fail("dqnhip_update_async_n: n must be >= 0");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->cfg.dp_world > 1 || h->dp_half || h->dp_shard) return fail("dqnhip_update_async_n: data-parallel learners use dqnhip_dp_update");
  if (h->next_phase != 0) return fail("dqnhip_update_async_n: a phased update is in progress (next phase %d)", h->next_phase);
  RingUse ring_use(h);
  RC(sync_dirty16(h));
  if (RO(h)->h_size < 1) RC(refresh_ring(h));
  if (RO(h)->h_size < 1) return fail("replay memory is empty");
  if (h->cfg.use_graph && !h->timing && !h->graph_failed) {
    if (n >= kMultiU && !h->graph_exec[4] && capture_graph(h, 4)) h->graph_failed = true;
    while (n >= kMultiU && h->graph_exec[4]) {
      HIPCHK(hipGraphLaunch(h->graph_exec[4], h->stream));
      h->h_actor_iter += kMultiU; h->h_critic_iter += kMultiU; n -= kMultiU;
    }
    // the remainder as graphs of 8 / 4 / 2 updates (the same launch sequence as the sixteen-update graph, cut shorter), then one
    for (int k = 0; k < 3 && !h->graph_failed; ++k) {
      const int sz = 8 >> k;
      if (n < sz) continue;
      if (!h->graph_small[k] && capture_graph(h, 4, nullptr, sz, &h->graph_small[k])) { h->graph_failed = true; break; }
      HIPCHK(hipGraphLaunch(h->graph_small[k], h->stream));
      h->h_actor_iter += sz; h->h_critic_iter += sz; n -= sz;
    }
    if (n > 0 && !h->graph_failed && !h->graph_exec[0] && capture_graph(h, 0)) h->graph_failed = true;
    while (n > 0 && h->graph_exec[0]) {
      HIPCHK(hipGraphLaunch(h->graph_exec[0], h->stream));
      h->h_actor_iter += 1; h->h_critic_iter += 1; n -= 1;
    }
  }
  for (; n > 0; --n)
    for (int p = 0; p < 3; ++p) RC(run_phase(h, p, nullptr));
  return 0;
}

int dqnhip_update_phase(dqnhip_handle h, int32_t phase, const int32_t* idx_host) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  HIPCHK(hipSetDevice(h->cfg.device));
  // (with DQNHIP_DP_HALF_GRADS the exchange — bf16 image, all-reduce, widening by the clip-norm pass — lives inside
  // dqnhip_dp_update: a caller-driven exchange between phases would leave phase 1 / 2 reading a stale bf16 image)
  if (h->dp_half) return fail("dqnhip_update_phase: this learner exchanges bf16 gradients (DQNHIP_DP_HALF_GRADS): use dqnhip_dp_update");
  if (h->dp_shard) return fail("dqnhip_update_phase: this learner's optimiser is sharded over its group (DQNHIP_DP_SHARD_OPT): use dqnhip_dp_update");
  // 0 -> 1 -> 2 or 10 -> 11 -> 1 -> 2: a phase run out of order would apply stale gradients
  // and advance the iteration counters
  const int expect = h->next_phase;
  const bool ok = (phase == 0 || phase == 10) ? (expect == 0) : (phase == expect);
  if (!ok) return fail("dqnhip_update_phase: phase %d out of order (expected %s)", phase,
                       expect == 0 ? "0 or 10" : expect == 1 ? "1" : expect == 2 ? "2" : "11");
  const int* idx_dev = nullptr;
  int rc;
  if (phase != 0 && phase != 10) rc = run_phase(h, phase, idx_dev);
  else {
    RingUse ring_use(h);
    rc = sync_dirty16(h);
    if (!rc) rc = stage_indices(h, idx_host, &idx_dev);
    if (!rc) rc = run_phase(h, phase, idx_dev);
  }
  // a failed phase abandons the update: the next one starts from phase 0 / 10 again (one transient error must not
  // wedge the learner in "out of order" for good)
  h->next_phase = rc ? 0 : (phase == 0 ? 1 : phase == 10 ? 11 : phase == 11 ? 1 : phase == 1 ? 2 : 0);
  return rc;
}

int dqnhip_update_abort(dqnhip_handle h) {
  if (!h) return fail("null handle");
  h->next_phase = 0;
  return 0;
}

int dqnhip_read_stats(dqnhip_handle h, float* critic_loss, float* avg_q) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  // {critic_loss, avg_q, flags}: the last block of every update writes them into this pinned, host-mapped buffer itself
  // (tick_body): no device-to-host copy, the stream sync is the only wait
  HIPCHK(hipStreamSynchronize(h->stream));
  if (critic_loss) *critic_loss = h->pinned_stats[0];
  if (avg_q) *avg_q = h->pinned_stats[1];
  int flags = 0; memcpy(&flags, &h->pinned_stats[2], sizeof flags);
  if (flags) { HIPCHK(hipMemsetAsync(&h->st->flags, 0, sizeof(int), h->stream)); h->pinned_stats[2] = 0.0f; }   // sticky until reported
  // CHECK(std::isfinite(target)) (src/dqn.cpp:898) and CHECK(std::isfinite(critic_loss)) (:906) — the
  // reference aborts; here: an error code from the first read after the offending update, whichever
  // entry point (blocking, async, phased, hipGraph) ran it
  if (flags & kFlagTarget) return fail("Target not finite!");
  if (flags & kFlagGradNorm) return fail("Gradient norm not finite: the clip+Adam step was skipped (fp16: lower cfg.loss_scale)");
  if (!std::isfinite(h->pinned_stats[0])) return fail("Critic loss not finite!");
  return 0;
}

int dqnhip_skipped_steps(dqnhip_handle h, int64_t* count) {
  if (!h || !count) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  int v = 0;
  HIPCHK(hipMemcpyAsync(h->pinned_stats + 8, &h->st->skipped_steps, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(&v, h->pinned_stats + 8, sizeof v);
  *count = v;
  return 0;
}

int dqnhip_update(dqnhip_handle h, const int32_t* idx_host, float* critic_loss, float* avg_q) {
  RC(dqnhip_update_async(h, idx_host));
  return dqnhip_read_stats(h, critic_loss, avg_q);
}

// dqnhip_update with the NEXT update's indices known one call ahead (see dqnhip.h).  The reference's driver runs its updates in
// bursts (src/dqn_main.cpp:359-361: `for (i < n_updates) dqn->Update()`), one blocking call at a time; with idx_next the burst gets the
// schedule of a multi-update graph — the next update's gather in this update's critic optimiser launch, its four first layers in the
// actor's — one graph launch per update.  What rode along is used only if the next call's idx equals idx_next and nothing changed
// weights, iteration counters or the replay memory in between; otherwise that call starts a fresh chain (own gather, own first
// layers: what the riders left is overwritten).  Every update computes exactly what dqnhip_update computes on the same indices.
int dqnhip_update_chained(dqnhip_handle h, const int32_t* idx_host, const int32_t* idx_next, float* critic_loss, float* avg_q) {
  if (!h) return fail("null handle");
  if (!idx_host) return fail("dqnhip_update_chained: explicit indices required (on-device sampling: dqnhip_update_async_n)");
  HIPCHK(hipSetDevice(h->cfg.device));
  const UpdatePlan P = plan_of(h);
  const bool can = P.early_l0 && !P.dp && !h->comm && h->cfg.use_graph && !h->timing && !h->graph_failed && h->sharers == 0 && h->w_owner == nullptr;
  if (!can) { h->chain_valid = false; return dqnhip_update(h, idx_host, critic_loss, avg_q); }
  if (h->next_phase != 0) return fail("dqnhip_update_chained: a phased update is in progress (next phase %d)", h->next_phase);
  {
    RingUse ring_use(h);
    RC(refresh_ring(h));
    const long long size = RO(h)->h_size;
    if (size < 1) return fail("replay memory is empty");
    for (int i = 0; i < h->B; ++i) {
      if (idx_host[i] < 0 || idx_host[i] >= size) return fail("sampled index %d = %d out of range [0,%lld)", i, idx_host[i], size);
      if (idx_next && (idx_next[i] < 0 || idx_next[i] >= size)) return fail("next sampled index %d = %d out of range [0,%lld)", i, idx_next[i], size);
    }
    if (!h->idx_next_pinned[0])
      for (int i = 0; i < 2; ++i) {
        HIPCHK(hipHostMalloc((void**)&h->idx_next_pinned[i], h->B * sizeof(int), hipHostMallocMapped));
        memset(h->idx_next_pinned[i], 0, h->B * sizeof(int));
        void* d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, h->idx_next_pinned[i], 0)); h->idx_next_dev[i] = (const int*)d;
      }
    HIPCHK(hipStreamSynchronize(h->stream));       // the pinned index buffers may still be in flight from an earlier (asynchronous) update
    const bool cont = h->chain_valid && h->chain_epoch == h->epoch && h->chain_ring_epoch == RO(h)->epoch &&
                      memcmp(h->chain_idx.data(), idx_host, h->B * sizeof(int32_t)) == 0;
    const int par = cont ? h->chain_par : 0;         // this update's panel parity
    const int which = cont ? (par ? 6 : 7) : 5;
    if (!cont) memcpy(h->idx_pinned, idx_host, h->B * sizeof(int));
    h->chain_valid = false;
    if (idx_next) {
      memcpy(h->idx_next_pinned[par ^ 1], idx_next, h->B * sizeof(int));
      h->chain_idx.assign(idx_next, idx_next + h->B);
    }
    if (!h->graph_exec[which] && capture_graph(h, which)) { h->graph_failed = true; return dqnhip_update(h, idx_host, critic_loss, avg_q); }
    HIPCHK(hipGraphLaunch(h->graph_exec[which], h->stream));
    h->h_actor_iter += 1; h->h_critic_iter += 1;
    h->epoch += 1;
    if (idx_next) { h->chain_valid = true; h->chain_par = par ^ 1; h->chain_epoch = h->epoch; h->chain_ring_epoch = RO(h)->epoch; }
  }
  // (both null: enqueue only — the caller does its own work while the update runs, e.g. drawing the prediction after next, and
  // collects the scalars with dqnhip_read_stats)
  if (critic_loss == nullptr && avg_q == nullptr) return 0;
  return dqnhip_read_stats(h, critic_loss, avg_q);
}

// One-deep pipelined form of dqnhip_update: enqueues update t and returns the scalars of update t-1 (zeros on the
// first call).  The host then waits for update t-1 only, while update t is already queued behind it — the device
// never idles on the host's index draw, the H2D of the indices or the read-back, which dqnhip_update pays on
// every call.  Indices and scalars use two pinned slots each (nothing in flight is overwritten).
int dqnhip_update_pipelined(dqnhip_handle h, const int32_t* idx_host, float* critic_loss, float* avg_q) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->cfg.dp_world > 1 || h->dp_half || h->dp_shard) return fail("dqnhip_update_pipelined: data-parallel learners use dqnhip_update_phase / dqnhip_dp_update");
  if (h->next_phase != 0) return fail("dqnhip_update_pipelined: a phased update is in progress (next phase %d)", h->next_phase);
  if (!h->pipe_ev[0]) {
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipEventCreateWithFlags(&h->pipe_ev[i], hipEventDisableTiming));
      HIPCHK(hipHostMalloc((void**)&h->pipe_idx_pinned[i], h->B * sizeof(int), hipHostMallocMapped));
      { void* d = nullptr; HIPCHK(hipHostGetDevicePointer(&d, h->pipe_idx_pinned[i], 0)); h->pipe_idx_dev[i] = (int*)d; }
      HIPCHK(hipHostMalloc((void**)&h->pipe_stats[i], 64, hipHostMallocDefault));
      memset(h->pipe_stats[i], 0, 64);
    }
  }
  const int slot = (int)(h->pipe_count & 1);
  {
    RingUse ring_use(h);
    RC(sync_dirty16(h));
    const int* idx_dev = nullptr;
    if (idx_host) {
      RC(refresh_ring(h));
      for (int i = 0; i < h->B; ++i)
        if (idx_host[i] < 0 || idx_host[i] >= RO(h)->h_size) return fail("sampled index %d = %d out of range [0,%lld)", i, idx_host[i], RO(h)->h_size);
      // slot's previous user was update t-2, whose completion the previous call already waited for
      memcpy(h->pipe_idx_pinned[slot], idx_host, h->B * sizeof(int));
      idx_dev = h->pipe_idx_dev[slot];      // device alias of the pinned slot (no H2D copy)
    } else if (RO(h)->h_size < 1) {
      RC(refresh_ring(h));
      if (RO(h)->h_size < 1) return fail("replay memory is empty");
    }
    const int which = idx_host ? 2 + slot : 0;
    if (h->cfg.use_graph && !h->timing && !h->graph_failed) {
      if (!h->graph_exec[which] && capture_graph(h, which, idx_dev)) h->graph_failed = true;
    }
    if (h->cfg.use_graph && !h->timing && !h->graph_failed && h->graph_exec[which]) {
      HIPCHK(hipGraphLaunch(h->graph_exec[which], h->stream));
      h->h_actor_iter += 1; h->h_critic_iter += 1;
    } else {
      for (int p = 0; p < 3; ++p) RC(run_phase(h, p, idx_dev));
    }
  }
  HIPCHK(hipMemcpyAsync(h->pipe_stats[slot], &h->st->critic_loss, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipEventRecord(h->pipe_ev[slot], h->stream));
  const int prev = slot ^ 1;
  float loss = 0.f, q = 0.f; int flags = 0;
  if (h->pipe_count > 0) {
    HIPCHK(hipEventSynchronize(h->pipe_ev[prev]));
    loss = h->pipe_stats[prev][0]; q = h->pipe_stats[prev][1];
    memcpy(&flags, &h->pipe_stats[prev][2], sizeof flags);
  }
  h->pipe_count += 1;
  if (critic_loss) *critic_loss = loss;
  if (avg_q) *avg_q = q;
  if (flags) {     // sticky on the device: report through the blocking path, which clears them
    float l2, q2;
    const int rc = dqnhip_read_stats(h, &l2, &q2);      // syncs the stream: update t (enqueued above) has completed too
    // update t's read-back into the other slot was enqueued BEFORE the flags were cleared and still carries them: this
    // report covers it, so the next call must not raise the same flag again
    memset(&h->pipe_stats[slot][2], 0, sizeof(float));
    return rc ? 1 : fail("update flags raised");
  }
  if (!std::isfinite(loss)) return fail("Critic loss not finite!");
  return 0;
}

// Solver::ApplyUpdate() of one net in isolation (actor_solver_->ApplyUpdate(), src/dqn.cpp:964; the tail of
// critic_solver_->Step(1), :904) on the gradient currently in the net's arena (e.g. dqnhip_set_params(KIND_G)):
// ClipGradients + Adam + Net::Update + the soft update of that net's target under the condition of :967, then
// set_iter(iter + 1) of that solver (:965).  The clip norm is taken from the arena itself (k_sumsq), as after an
// all-reduce; the same k_adam_soft pass as inside an update.
int dqnhip_apply_update(dqnhip_handle h, int32_t net) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  if (h->next_phase != 0) return fail("dqnhip_apply_update: a phased update is in progress (next phase %d)", h->next_phase);
  HIPCHK(hipSetDevice(h->cfg.device));
  RC(sync_dirty16(h));
  const NetLayout& l = layout_of(h, net);
  hipLaunchKernelGGL(k_sumsq, dim3(h->n_part_dp), dim3(256), 0, h->stream, h->g[net], l.arena / 4, h->part_dp);
  HIPCHK(hipGetLastError());
  RC(adam_launch(h, h->stream, net, h->part_dp, h->n_part_dp, 0, l.arena, nullptr, false));   // no gather ran: the pass evaluates its own correction
  hipLaunchKernelGGL(k_advance_iter, dim3(1), dim3(1), 0, h->stream, h->st, (int)net, h->stats_dev);
  HIPCHK(hipGetLastError());
  if (net == DQNHIP_ACTOR) h->h_actor_iter += 1; else h->h_critic_iter += 1;
  return 0;
}

// Solver::ApplyUpdate() of one net evaluated the way a `world`-rank group with a SHARDED optimiser evaluates it
// (DQNHIP_DP_SHARD_OPT), by this one learner standing in for every rank in turn: per slice r the sum of squares of the
// gradient's floats [r, r + 1) * arena / world (k_sumsq + k_shard_scal — each rank's share of the clip norm), their sum in
// rank order (what the 4-float all-reduce leaves on every rank), then clip + Adam + Net::Update + soft update on slice r with
// that norm (k_adam_soft on the sub-range), then set_iter(iter + 1).  No exchange is needed because the gradient in the
// arena already IS the reduced one.  world = 1 is dqnhip_apply_update bit for bit; world > 1 differs from it only through
// the order in which the clip norm is summed (identical bits whenever the clip is inactive).
int dqnhip_apply_update_sharded(dqnhip_handle h, int32_t net, int32_t world) {
  if (!h) return fail("null handle");
  h->epoch += 1;
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  if (h->next_phase != 0) return fail("dqnhip_apply_update_sharded: a phased update is in progress (next phase %d)", h->next_phase);
  const NetLayout& l = layout_of(h, net);
  if (world < 1 || l.arena % ((size_t)4 * world)) return fail("apply_update_sharded: the arena (%zu floats) must be divisible by 4 x world = %d", l.arena, 4 * world);
  HIPCHK(hipSetDevice(h->cfg.device));
  RC(sync_dirty16(h));
  if (!h->shard_total) HIPCHK(hipMalloc(&h->shard_total, 8 * sizeof(float)));
  const size_t slice = l.arena / (size_t)world;
  float* tail = h->shard_total + 4;
  for (int r = 0; r < world; ++r) {
    hipLaunchKernelGGL(k_sumsq, dim3(h->n_part_dp), dim3(256), 0, h->stream, h->g[net] + r * slice, slice / 4, h->part_dp);
    hipLaunchKernelGGL(k_shard_scal, dim3(1), dim3(256), 0, h->stream, (const float*)h->part_dp, h->n_part_dp, tail);
    hipLaunchKernelGGL(k_shard_accumulate, dim3(1), dim3(1), 0, h->stream, h->shard_total, (const float*)tail, r == 0 ? 1 : 0);
    HIPCHK(hipGetLastError());
  }
  for (int r = 0; r < world; ++r) RC(adam_launch(h, h->stream, net, h->shard_total, 1, r * slice, (r + 1) * slice, nullptr, false));
  hipLaunchKernelGGL(k_advance_iter, dim3(1), dim3(1), 0, h->stream, h->st, (int)net, h->stats_dev);
  HIPCHK(hipGetLastError());
  if (net == DQNHIP_ACTOR) h->h_actor_iter += 1; else h->h_critic_iter += 1;
  return 0;
}

int dqnhip_grad_buffer(dqnhip_handle h, int32_t net, void** dptr, size_t* nfloats) {
  if (!h) return fail("null handle");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  if (dptr) *dptr = h->g[net];
  if (nfloats) *nfloats = layout_of(h, net).arena + 4;
  return 0;
}

int dqnhip_benchmark(dqnhip_handle h, int32_t warmup, int32_t iterations, float* avg_ms) {
  if (!h) return fail("null handle");
  if (iterations < 1) return fail("iterations must be >= 1");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (warmup > 0) RC(dqnhip_update_async_n(h, warmup));
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipEventRecord(a, h->stream));
  RC(dqnhip_update_async_n(h, iterations));
  HIPCHK(hipEventRecord(b, h->stream));
  HIPCHK(hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a); hipEventDestroy(b);
  if (avg_ms) *avg_ms = ms / iterations;
  return 0;
}


// What the reference's driver gets from this library (src/dqn_main.cpp:361 -> DQN::Update -> UpdateActorCritic, and
// DQN::Benchmark, src/dqn.cpp:487-498, which loops over it): host-drawn indices (std::mt19937 +
// uniform_int_distribution, SampleTransitionsFromMemory :501-509), staged to the device, and a BLOCKING read of
// (critic_loss, avg_q) after every update.  pipelined != 0: dqnhip_update_pipelined instead.
int dqnhip_benchmark_blocking(dqnhip_handle h, int32_t warmup, int32_t iterations, uint64_t seed, int32_t pipelined, float* avg_ms) {
  if (!h) return fail("null handle");
  if (iterations < 1) return fail("iterations must be >= 1");
  HIPCHK(hipSetDevice(h->cfg.device));
  int32_t size = 0;
  RC(dqnhip_memory_size(h, &size));
  if (size < 1) return fail("replay memory is empty");
  std::mt19937 rng((uint32_t)seed);
  std::vector<int32_t> idx(h->B);
  float loss = 0, avgq = 0;
  std::vector<int32_t> nxt(h->B);
  bool have_next = false;
  auto one = [&]() -> int {
    if (pipelined == 2) {
      // the drop-in's chained form (dqn_dropin.cpp UpdateActorCritic): the next update's indices are drawn one call ahead
      // (drawing the prediction after next while the update runs — an enqueue-only call, then dqnhip_read_stats — was measured
      // too: 3 334-3 350 against 3 335-3 347 updates/s, the draw of 256 indices is ~1 us; not kept)
      if (have_next) idx.swap(nxt); else for (int32_t& i : idx) i = std::uniform_int_distribution<int>(0, size - 1)(rng);
      for (int32_t& i : nxt) i = std::uniform_int_distribution<int>(0, size - 1)(rng);
      have_next = true;
      return dqnhip_update_chained(h, idx.data(), nxt.data(), &loss, &avgq);
    }
    for (int32_t& i : idx) i = std::uniform_int_distribution<int>(0, size - 1)(rng);
    return pipelined ? dqnhip_update_pipelined(h, idx.data(), &loss, &avgq) : dqnhip_update(h, idx.data(), &loss, &avgq);
  };
  for (int i = 0; i < warmup; ++i) RC(one());
  HIPCHK(hipStreamSynchronize(h->stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iterations; ++i) RC(one());
  HIPCHK(hipStreamSynchronize(h->stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (pipelined == 1) RC(dqnhip_read_stats(h, &loss, &avgq));      // drains the one outstanding read-back
  if (avg_ms) *avg_ms = (float)(ms / iterations);
  return 0;
}

}  // extern "C"
