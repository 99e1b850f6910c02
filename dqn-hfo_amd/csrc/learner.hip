// learner.hip — the device-resident actor-critic learner behind include/dqnhip.h.
//
// Host-side orchestration of DQN::UpdateActorCritic (reference src/dqn.cpp:828-972)
// as a fixed sequence of gfx950 kernels on one HIP stream with no host sync:
// nothing crosses PCIe per update except (optionally) B sampled indices in and
// two floats out.  See DESIGN.md for the data layout and the kernel list.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <zlib.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <mutex>
#include <chrono>
#include <thread>

#include "../../include/dqnhip.h"
#include "../../include/dqnhip_env.h"
#include "env.hip.h"
#include "gemm_direct.hip.h"
#include "hgemm.hip.h"
#include "small_kernels.hip.h"

using namespace dqnhip;

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return 1;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)
#define RC(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_z(size_t x, size_t m) { return (x + m - 1) / m * m; }

constexpr int kMaxL = DQNHIP_MAX_HIDDEN;
// fp16 learner: ALL wgrads of a net in one launch of 128x128 tiles from this many minibatch rows (the reduction
// length); below, the per-layer form (a layer's dgrad + wgrad sharing a launch of 64x64 split-K tiles) is as fast or
// faster (measured at 256 / 512 / 1024 / 2048 / 4096 rows: +0.7 / -6 / -28 / -37 / -68 us per update)
constexpr int kGroupMinRows = 512;

// Internal parameter arena of one net: tower layer l has W_l[dims[l+1]][kp[l]] (K
// padded to a multiple of 64 so every GEMM tile is whole) and b_l[dims[l+1]];
// the head(s) are stored as one [NH][H] matrix (action_layer rows 0-3,
// actionpara_layer rows 4-9: the Split layer disappears) and bh[16].
struct NetLayout {
  int L = 0, in_dim = 0, NH = 0;
  int dims[kMaxL + 1] = {0};   // logical widths: dims[0] = in_dim
  int kp[kMaxL + 1] = {0};     // padded widths of each activation panel
  size_t w_off[kMaxL] = {0}, b_off[kMaxL] = {0}, hw_off = 0, hb_off = 0;
  size_t arena = 0;            // floats, multiple of 64
  size_t dense = 0;            // dense (Caffe-order) parameter count
  // sum-of-squares partial slots
  int part_off[kMaxL + 1] = {0};  // per tower layer, then head
  int part_db = 0;                // fp16 learner: first slot of the bias-gradient workgroups
  int n_part = 0;
};

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

struct TimingRec { int family; hipEvent_t a, b; };

}  // namespace

struct dqnhip_learner {
  dqnhip_config cfg;
  int B = 0, S = 0, L = 0;
  NetLayout la, lc;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // parameter arenas
  float* w[4] = {nullptr, nullptr, nullptr, nullptr};
  float* m[2] = {nullptr, nullptr};
  float* v[2] = {nullptr, nullptr};
  float* g[2] = {nullptr, nullptr};     // inside grad_base
  float* grad_base = nullptr; bool own_grad = false;
  // replay
  Ring ring{};
  DevState* st = nullptr;
  int* done_counter = nullptr;
  long long h_head = 0, h_size = 0;     // host mirror of (head,size)
  bool ring_stale = false;              // the device changed (head,size) on its own (env front-end)
  // sharing (DQN::ShareParameters / ShareReplayMemory, src/dqn.cpp:1036-1083): a sharer keeps
  // its own allocations and reads the owner's through these
  dqnhip_learner* ring_owner = nullptr; // whose ring / (head,size) this learner uses (nullptr: own)
  dqnhip_learner* w_owner = nullptr;    // owner of the shared first layers
  size_t shared_fl[2] = {0, 0};         // arena floats [0, shared_fl) of actor / critic (+targets) live in w_owner
  int sharers = 0;                      // learners that reference this one
  bool ring_shared = false;             // more than one learner uses this ring: order users across streams
  hipEvent_t ring_ev = nullptr; hipStream_t ring_last = nullptr; bool ring_ev_valid = false;
  std::mutex ring_mu;
  int h_actor_iter = 0, h_critic_iter = 0;
  unsigned long long sample_states_calls = 0;
  // native data parallelism (dqnhip_dp_*): one RCCL communicator per learner
  ncclComm_t comm = nullptr;
  bool dp_per_layer = false;            // bucket the gradient all-reduce per layer on comm_stream
  bool dp_half = false;                 // gradients cross the links as bf16 (half the bytes); [loss, q, flag] tails stay fp32
  // DQNHIP_DP_SHARD_OPT: reduce-scatter -> clip + Adam + soft update on this rank's 1/N slice of each arena -> all-gather of
  // the updated online and target weights (m, v of the other slices go stale until dqnhip_dp_gather_state)
  bool dp_shard = false;
  float* shard_total = nullptr;         // dqnhip_apply_update_sharded: the group's sum of squares, accumulated rank by rank
  bool shard_stale = false;             // a sharded update ran since the last dqnhip_dp_gather_state: m, v of foreign slices are stale
  uint16_t* g16[2] = {nullptr, nullptr};   // bf16 transfer image of each gradient arena (dp_half)
  float* dp_tails = nullptr;            // dp_half: {critic tail[4], actor tail[4]}, one fp32 all-reduce with the actor's gradients
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev[2] = {nullptr, nullptr};
  hipGraphExec_t dp_graph = nullptr;    // the whole data-parallel update (collectives included), captured
  bool dp_graph_failed = false;
  hipGraphExec_t dp_graph_n = nullptr;  // kMultiU of them (dqnhip_dp_update_n)
  bool dp_graph_n_failed = false;
  int next_phase = 0;                   // dqnhip_update_phase order check (0: an update may start)
  // minibatch panels / activations: pass 0 AT, 1 A, 2 CT, 3 C1, 4 C2
  float* Xa_s = nullptr; float* Xa_n = nullptr; float* Xc_tr = nullptr; float* Xc_pl = nullptr; float* Xc_nx = nullptr;
  float* act[5][kMaxL + 1] = {{nullptr}};
  float* dZa[kMaxL + 1] = {nullptr};
  float* dZc[kMaxL + 1] = {nullptr};
  float *mb_reward = nullptr, *mb_mc = nullptr, *mb_term = nullptr;
  int* mb_idx = nullptr;
  int* idx_pinned = nullptr;
  const int* idx_pinned_dev = nullptr;  // device alias of idx_pinned: the gather reads explicit indices straight from host memory (no H2D copy)
  float* stats_dev = nullptr;           // device alias of pinned_stats: the update's last block writes {loss, avg_q, flags} there
  float *aout_t16 = nullptr, *aout16 = nullptr, *dA16 = nullptr;
  float *q_t = nullptr, *q1 = nullptr, *q2 = nullptr, *y = nullptr, *dq = nullptr;
  float* loss_partial = nullptr; double* q_partial = nullptr; int n_head_blocks = 0;
  float* part[2] = {nullptr, nullptr};  // GEMM-epilogue sumsq partials per net
  float* part_dp = nullptr; int n_part_dp = 0;
  float* head_slab = nullptr; int* head_ticket = nullptr;   // k_head_bwd cross-block reduction
  float* head_slab2 = nullptr;                               // k_head_bwd_big row-chunk slabs (minibatch >= 1024)
  // mixed precision (cfg.precision == DQNHIP_FP16): ONE fp16 mirror of each weight arena and batch-major fp16
  // activation / gradient panels; the dgrad and wgrad GEMMs read them reduction-major (hgemm.hip.h), so no
  // transposed copy of anything exists
  bool fp16 = false;
  float ls_c = 1.f, ls_q = 1.f, ls_a = 1.f;  // static loss scales: critic step, dQ/da pass, actor step
  int k16[2][kMaxL + 1] = {{0}};             // fp16 panel widths per net kind (k16[.][0] = in_dim rounded to 128)
  h16* w16a[4] = {nullptr, nullptr, nullptr, nullptr};   // fp16 mirror of each weight arena (written by the Adam pass)
  h16* w16[4][kMaxL] = {{nullptr}};          // = w16a[net] + w_off[i]: [N_out][kp]
  h16* act16[5][kMaxL + 1] = {{nullptr}};    // [B][k16]
  h16* dZ16[2][kMaxL + 1] = {{nullptr}};     // [B][k16]   per net kind
  bool w16_dirty[4] = {true, true, true, true};
  std::vector<void*> allocs16;
  // host-staging for add_transitions / acting
  void* stage_dev = nullptr; size_t stage_bytes = 0;
  float* act_buf = nullptr; size_t act_floats = 0;
  float* pinned_stats = nullptr;
  // timing
  bool timing = false;
  std::vector<TimingRec> recs;
  // graph
  int cap_u = -1;              // while capturing a multi-update graph: the position of the update being captured (else -1)
  hipGraphExec_t graph_exec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0]: device-sampled, [1]: explicit idx (pinned buffer -> memcpy node), [2], [3]: explicit idx in the pipelined slots, [4]: kMultiU device-sampled updates (dqnhip_update_async_n)
  bool graph_failed = false;
  // dqnhip_update_pipelined
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};
  int* pipe_idx_dev[2] = {nullptr, nullptr}; int* pipe_idx_pinned[2] = {nullptr, nullptr};
  float* pipe_stats[2] = {nullptr, nullptr};
  unsigned long long pipe_count = 0;
};

namespace {

using H = dqnhip_learner;

// ring owner / weight views under sharing
inline H* RO(H* h) { return h->ring_owner ? h->ring_owner : h; }
inline const H* RO(const H* h) { return h->ring_owner ? h->ring_owner : h; }
inline float* wat(const H* h, int net, size_t off) {
  return ((off < h->shared_fl[net & 1]) ? h->w_owner->w[net] : h->w[net]) + off;
}

// Orders the users of a SHARED ring across their streams in host-call order: each user waits
// for the previous user's completion event.  No-op (no lock, no event) for a private ring.
struct RingUse {
  H* o; hipStream_t st; bool on;
  RingUse(H* h) : o(RO(h)), st(h->stream), on(RO(h)->ring_shared) {
    if (!on) return;
    o->ring_mu.lock();
    if (o->ring_ev_valid && o->ring_last != st) hipStreamWaitEvent(st, o->ring_ev, 0);
  }
  ~RingUse() {
    if (!on) return;
    hipEventRecord(o->ring_ev, st); o->ring_last = st; o->ring_ev_valid = true;
    o->ring_mu.unlock();
  }
};

const char* kFamily[] = {"gemm_fwd_lds_4x2", "gemm_dgrad", "gemm_wgrad", "adam", "gemm_bwd_pair", "gemm_fwd_lds_2x2", "gemm_fwd_direct",
                         "hgemm_fwd", "hgemm_dgrad", "hgemm_wgrad"};
constexpr int kNumFamily = 10;

// Timing mode: the NEXT kernel launch (through direct_launch / adam_launch) is bracketed by the
// dispatch packet's own timestamps (hipExtLaunchKernelGGL start/stop events).
struct ScopedTiming {
  ScopedTiming(H* h, int fam, hipStream_t) {
    if (h->timing) {
      hipEvent_t a = nullptr, b = nullptr;
      hipEventCreate(&a); hipEventCreate(&b);
      launch_timer().start = a; launch_timer().stop = b;
      h->recs.push_back({fam, a, b});
    }
  }
};

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

const NetLayout& layout_of(const H* h, int net) { return (net & 1) ? h->lc : h->la; }

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

size_t grad_arena_floats(const NetLayout& la, const NetLayout& lc) { return la.arena + 64 + lc.arena + 64; }

// ---- forward / backward building blocks ----------------------------------------

// seed_w / seed_out: the TOP layer's launch also writes the dq = -1 pass's tower-top gradient (GemmProblem::seed_w)
struct FwdPass { int net; const NetLayout* l; float** act; const float* seed_w = nullptr; float* seed_out = nullptr; };

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
int tower_forward(H* h, hipStream_t st, const FwdPass* passes, int n, int rows) {
  for (int i = 0; i < passes[0].l->L; ++i) RC(layer_forward(h, st, passes, n, rows, i));
  return 0;
}

// Tower backward from dZ[L] (gradient wrt the last tower pre-activation) down, one launch per layer on `st`.
// want_w: produce dW/db (+sumsq partials) into garena; input_grad: also dZ[0].
// in_lo / in_hi: when only these input columns of dZ[0] are consumed (the critic's action columns), the
// first layer's dgrad computes just the 16-column tiles that cover them.
// RCCL sum all-reduce of one slice of a gradient arena on the communication stream, ordered after
// everything enqueued on `st` so far (per-layer bucketing; defined with dqnhip_dp_*)
int dp_reduce_slice(H* h, hipStream_t st, int net, size_t off, size_t count);

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
int tower_backward(H* h, hipStream_t st, const NetLayout& l, int net, float* garena, float* partial,
                   float** act, float** dZ, int rows, bool want_w, bool input_grad, int in_lo = 0, int in_hi = -1,
                   const HeadWgradRider* rider = nullptr, const QHeadRider* qrider = nullptr) {
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
  bool shifted = want_w && !input_grad && l.L >= 2 && rows % 16 == 0 && !(h->cfg.tuning_flags & DQNHIP_TUNE_BWD_UNSHIFTED);
  for (int i = 1; i < l.L && shifted; ++i) shifted = !bwd_layer_is_pair(l, i, rows) && l.kp[i] % 64 == 0 && l.dims[i + 1] % 64 == 0;
  if (shifted) {
    {
      GemmBatch bd{}; bd.n = 1; bd.prob[0] = dgrad_of(l.L - 1);
      ScopedTiming t(h, 1, st);
      if (lds_ok_of(l.L - 1)) HIPCHK((dgrad_lds_launch<1, 1>(bd, st))); else HIPCHK((dgrad_direct_launch<1, 1>(bd, st)));
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
      if (l.NH == 1) HIPCHK((wgrad_tail_launch<1>(b, rider ? *rider : none, st))); else HIPCHK((wgrad_tail_launch<kNO>(b, rider ? *rider : none, st)));
      if (h->comm && h->dp_per_layer) RC(dp_reduce_slice(h, st, net, l.w_off[0], layer_slice(0) + layer_slice(1)));
    }
    if (qrider) return fail("internal: the q-head rider found no carrier launch");
    return 0;
  }
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

template <int NH, int MODE>
int head_forward(H* h, hipStream_t st, const HeadArgs& a, const HeadArgs* b = nullptr) {
  HeadArgs2 a2{}; a2.p[0] = a; if (b) a2.p[1] = *b;
  if (NH > 1 && a.rows >= 1024 && a.H <= 1024 && a.H % 4 == 0)   // (single-head: the block-per-row form measured faster, 6.6 vs 8.8 us)
    hipLaunchKernelGGL((k_head_fwd_rows<NH, MODE>), dim3(256, b ? 2 : 1), dim3(256), 0, st, a2);
  else
    hipLaunchKernelGGL((k_head_fwd<NH, MODE>), dim3(std::min(a.rows, 1024), b ? 2 : 1), dim3(256), 0, st, a2);
  HIPCHK(hipGetLastError());
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

constexpr int kMultiU = 16;    // updates per replay of the multi-update graph (dqnhip_update_async_n; see capture_graph)
// Philox key of SampleTransitionsFromMemory: cfg.seed on rank 0 (what oracle/c_oracle.philox_indices
// reproduces); data-parallel ranks get distinct streams from the SAME cfg.seed, so that the weight
// initialisation (also keyed by cfg.seed) stays identical across the group
inline uint64_t sample_key(const H* h) { return (uint64_t)h->cfg.seed + 0x9E3779B97F4A7C15ull * (uint64_t)h->cfg.dp_rank; }
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
    g.o = GatherOut{h->Xa_s, h->Xa_n, la.kp[0], h->Xc_tr, h->Xc_pl, h->Xc_nx, lc.kp[0], h->mb_reward, h->mb_mc, h->mb_term, h->mb_idx};
  const int slot = pos > 0 ? (pos & 1) : 0;
  g.corr = h->st->adam_corr[slot]; g.soft_now = &h->st->soft_now[slot];
  g.beta1 = h->cfg.momentum; g.beta2 = h->cfg.momentum2; g.soft_update_freq = h->cfg.soft_update_freq;
  g.ahead = pos > 0 ? pos : -1; g.store_base = pos == 0 ? 1 : 0;
  g.blocks = (h->B + 3) / 4 + 1;
  return g;
}

// clip + Adam + Net::Update + soft target update over arena floats [begin, end)
// corr_pre: the update's first launch (k_gather) has left this step's bias correction in DevState::adam_corr
int adam_launch(H* h, hipStream_t st, int net, const float* partial, int n_partial, size_t begin, size_t end, const TickArgs* tick = nullptr,
                bool corr_pre = true) {
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
  const int blocks = (int)std::min<size_t>((a.n4 + 255) / 256, (size_t)1536);
  if (tick && h->cap_u >= 0 && h->cap_u + 1 < kMultiU) {
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
int sumsq_launch(H* h, int net, size_t begin = 0, size_t end = 0) {
  const NetLayout& l = layout_of(h, net);
  if (end == 0) end = l.arena;
  if (h->dp_half) hipLaunchKernelGGL(k_sumsq_bf16, dim3(h->n_part_dp), dim3(256), 0, h->stream, (const uint16_t*)h->g16[net] + begin, h->g[net] + begin, (end - begin) / 4, h->part_dp);
  else hipLaunchKernelGGL(k_sumsq, dim3(h->n_part_dp), dim3(256), 0, h->stream, h->g[net] + begin, (end - begin) / 4, h->part_dp);
  HIPCHK(hipGetLastError());
  return 0;
}
// this rank's slice of a net's arena under the sharded optimiser: floats [lo, hi)
inline void shard_range(const H* h, int net, size_t& lo, size_t& hi, int rank = -1) {
  const size_t slice = layout_of(h, net).arena / (size_t)h->cfg.dp_world;
  const size_t r = (size_t)(rank < 0 ? h->cfg.dp_rank : rank);
  lo = r * slice; hi = lo + slice;
}
// the optimiser step of one net inside a data-parallel update (phase 1: critic, phase 2: actor + bookkeeping)
int dp_allgather_weights(H* h, int net);
int dp_optimiser_step(H* h, hipStream_t st, int net, float* tail, const TickArgs* tick) {
  const NetLayout& l = layout_of(h, net);
  if (h->dp_shard) {
    // the exchange left this rank's slice of the reduced gradient in place and the group's sum of squares in tail[3]
    size_t lo, hi; shard_range(h, net, lo, hi);
    RC(adam_launch(h, st, net, tail + 3, 1, lo, hi, tick));
    return dp_allgather_weights(h, net);
  }
  RC(sumsq_launch(h, net));
  return adam_launch(h, st, net, h->part_dp, h->n_part_dp, 0, l.arena, tick);
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
int tower_backward16(H* h, hipStream_t st, int net, int p, float* garena, float* dZ32_0, int rows,
                     bool want_w, bool input_grad, float ls, float* partial = nullptr) {
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
    HIPCHK(hgemm_group_db_launch(gws, l.L, true, db, db_blocks, st, e0, e1));
  } else if (!input_grad && hgemm_uses_small_tile(gws[0]) && gws[0].K % 128 == 0) {
    // per-layer form: the first layer's wgrad (few tiles, long reduction) carries the column sums
    HIPCHK(hgemm_group_db_launch(gws, 1, false, db, db_blocks, st, e0, e1));
  } else {
    if (!input_grad) HIPCHK(hgemm_launch(gws[0], st, 0, e0, e1));
    hipLaunchKernelGGL(k_db16_cols<0>, dim3(db_blocks), dim3(256), 0, st, db);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int run_phase16(H* h, int phase, const int* idx_dev) {
  const int B = h->B, L = h->L;
  const NetLayout &la = h->la, &lc = h->lc;
  const bool dp = h->cfg.dp_world > 1 || h->dp_half || h->dp_shard;     // (a one-rank group with bf16 exchange / a sharded optimiser runs the N-rank code path)
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
      hipLaunchKernelGGL(k_head_q_train, dim3((B + 3) / 4), dim3(256), 0, st, t);
      HIPCHK(hipGetLastError());
    }
    {
      HeadBwdArgs a{}; a.dyh = h->dq; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X416 = h->act16[3][L];
      a.H = Hc; a.rows = B; a.dZ = nullptr; a.dW = h->g[1] + lc.hw_off; a.db = h->g[1] + lc.hb_off;
      a.partial = h->part[1] + lc.part_off[L];
      if (head_big_ok(h, B, Hc)) { a.dZ = nullptr; RC(head_backward_big<1>(h, st, a, h->dZ16[1][L], h->ls_c)); }
      else { a.dZ16 = h->dZ16[1][L]; a.scale16 = h->ls_c; RC(head_backward<1>(h, st, a)); }
    }
    RC(tower_backward16(h, st, DQNHIP_CRITIC, 3, h->g[1], nullptr, B, true, false, h->ls_c, part16 ? h->part[1] : nullptr));
    if (dp) {
      hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, (const float*)h->loss_partial,
                         h->n_head_blocks, (const double*)nullptr, 0, inv_batch, critic_tail, (float*)nullptr, (const DevState*)h->st);
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
    const bool fused_seed = !(h->cfg.tuning_flags & DQNHIP_TUNE_SEPARATE_HEAD_SEED);
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
    RC(tower_backward16(h, st, DQNHIP_CRITIC, 4, nullptr, h->dZc[0], B, false, true, h->ls_q));
    {
      HeadBwdArgs a{}; a.dXc = h->dZc[0]; a.ldx = lc.kp[0]; a.S = h->S; a.aout16 = h->aout16; a.dA16 = h->dA16;
      a.W = wat(h, DQNHIP_ACTOR, la.hw_off); a.X416 = h->act16[1][L]; a.H = Hh; a.rows = B; a.dZ = nullptr;
      if (fused_seed) {
        a.q_bias = wat(h, DQNHIP_CRITIC, lc.hb_off); a.q_out = h->q2; a.qsum_partial = h->q_partial;
        a.qr_W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.qr_X416 = h->act16[4][L]; a.qr_H = Hc;
      }
      a.dW = h->g[0] + la.hw_off; a.db = h->g[0] + la.hb_off; a.partial = h->part[0] + la.part_off[L];
      if (head_big_ok(h, B, Hh)) { a.dZ = nullptr; RC(head_backward_big<kNO>(h, st, a, h->dZ16[0][L], h->ls_a)); }
      else { a.dZ16 = h->dZ16[0][L]; a.scale16 = h->ls_a; RC(head_backward<kNO>(h, st, a)); }
    }
    RC(tower_backward16(h, st, DQNHIP_ACTOR, 1, h->g[0], nullptr, B, true, false, h->ls_a, part16 ? h->part[0] : nullptr));
    if (dp) {
      hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, (const float*)nullptr, 0,
                         (const double*)h->q_partial, B, inv_batch, (float*)nullptr, actor_tail, (const DevState*)h->st);
      HIPCHK(hipGetLastError());
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
int run_phase(H* h, int phase, const int* idx_dev) {
  if (h->fp16) return run_phase16(h, phase, idx_dev);
  const int B = h->B, L = h->L;
  const NetLayout &la = h->la, &lc = h->lc;
  const bool dp = h->cfg.dp_world > 1 || h->dp_half || h->dp_shard;     // (a one-rank group with bf16 exchange / a sharded optimiser runs the N-rank code path)
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
    FwdPass cp[2] = {pCT, pC1};
    if (split) {
      // data-parallel overlap form: the online actor's forward (phase 11) is left out so that it can
      // run while the critic gradients are being all-reduced
      RC(tower_forward(h, st, &pAT, 1, B));
      RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT)));
    } else {
      // actor_target(s') [src/dqn.cpp:889-891] and actor(s) [:910-911, pre-update weights] layer by layer in one
      // launch each, then critic_target(s', mu'(s')) and the critic(s, a) train forward [:904] likewise
      FwdPass ap[2] = {pAT, pA};
      RC(tower_forward(h, st, ap, 2, B));
      RC((head_forward<kNO, HEAD_ACTOR>(h, st, hAT, &hA)));     // both actors' heads in one launch
    }
    RC(tower_forward(h, st, cp, 2, B));
    {
      HeadTrainArgs t{};
      t.Xt = h->act[2][L]; t.Wt = wat(h, DQNHIP_CRITIC_TARGET, lc.hw_off); t.bt = wat(h, DQNHIP_CRITIC_TARGET, lc.hb_off);
      t.X = h->act[3][L]; t.W = wat(h, DQNHIP_CRITIC, lc.hw_off); t.b = wat(h, DQNHIP_CRITIC, lc.hb_off);
      t.H = Hc; t.rows = B; t.reward = h->mb_reward; t.mc = h->mb_mc; t.term = h->mb_term;
      t.q_target = h->q_t; t.q = h->q1; t.y = h->y; t.dq = h->dq; t.loss_partial = h->loss_partial;
      t.gamma = h->cfg.gamma; t.beta = h->cfg.beta; t.inv_batch = inv_batch; t.st = h->st;
      // with the head's dW / db riding in the net's last backward launch, the head's dZ comes out of this launch too
      if (!head_big_ok(h, B, Hc) && head_wgrad_can_ride(lc, B)) t.dZ = h->dZc[L];
      hipLaunchKernelGGL(k_head_q_train, dim3((B + 3) / 4), dim3(256), 0, st, t);
      HIPCHK(hipGetLastError());
    }
    // critic backward (rest of Step(1)): head (dgrad + ReLU' + wgrad fused), then tower; wgrad
    // writes (beta=0) so ClearParamDiffs/ZeroGradParameters (src/dqn.cpp:63-78, 908-909) vanish
    {
      HeadBwdArgs a{}; a.dyh = h->dq; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X4 = h->act[3][L];
      a.H = Hc; a.rows = B; a.dZ = h->dZc[L]; a.dW = h->g[1] + lc.hw_off; a.db = h->g[1] + lc.hb_off;
      a.partial = h->part[1] + lc.part_off[L];
      // the head's own gradients ride in the net's last backward launch (the first layer's narrow wgrad)
      const bool ride = !head_big_ok(h, B, Hc) && head_wgrad_can_ride(lc, B);
      HeadWgradRider r{h->dq, 1, h->act[3][L], Hc, B, a.dW, a.db, a.partial, Hc / kRiderCW};
      if (!ride) RC(head_backward<1>(h, st, a));          // (riding: dZ came out of k_head_q_train, dW / db come from the rider)
      RC(tower_backward(h, st, lc, DQNHIP_CRITIC, h->g[1], h->part[1], h->act[3], h->dZc, B, true, false, 0, -1, ride ? &r : nullptr));
    }
    if (dp) {
      hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, (const float*)h->loss_partial,
                         h->n_head_blocks, (const double*)nullptr, 0, inv_batch, critic_tail, (float*)nullptr, (const DevState*)h->st);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  if (phase == 1) {
    // ClipGradients + Adam + Net::Update of the critic, soft update of critic_target fused (one pass)
    FwdPass pC2{DQNHIP_CRITIC, &lc, h->act[4]};
    h->act[4][0] = h->Xc_pl;
    if (dp) RC(dp_optimiser_step(h, st, 1, critic_tail, nullptr));
    else RC(adam_launch(h, st, 1, h->part[1], lc.n_part, 0, lc.arena));
    // The seed of BackwardFrom(q_values_layer) [:918-923] — q diff = -1 per row, taken through the head and the top
    // layer's ReLU, input gradient only (the reference's discarded critic dW, SURVEY a11, is never computed) — does not
    // depend on q: it comes out of the top tower layer's forward epilogue, and q(s, mu(s)) itself [:913-916], which only
    // the statistics read, rides in the chain's last launch.  DQNHIP_TUNE_SEPARATE_HEAD_SEED: the head-backward launch
    // that used to sit between the forward and the backward chain (same arithmetic, one launch more).
    const bool fused_seed = !(h->cfg.tuning_flags & DQNHIP_TUNE_SEPARATE_HEAD_SEED);
    if (fused_seed) { pC2.seed_w = wat(h, DQNHIP_CRITIC, lc.hw_off); pC2.seed_out = h->dZc[L]; }
    RC(tower_forward(h, st, &pC2, 1, B));                // critic(s, mu(s)), UPDATED weights [:913-916]
    const QHeadRider qr{h->act[4][L], wat(h, DQNHIP_CRITIC, lc.hw_off), wat(h, DQNHIP_CRITIC, lc.hb_off), h->q2, h->q_partial, Hc, B, (B + 3) / 4};
    if (!fused_seed) {
      HeadBwdArgs a{}; a.dyh = nullptr; a.lddy = 1; a.W = wat(h, DQNHIP_CRITIC, lc.hw_off); a.X4 = h->act[4][L];
      a.H = Hc; a.rows = B; a.dZ = h->dZc[L];
      a.q_bias = wat(h, DQNHIP_CRITIC, lc.hb_off); a.q_out = h->q2; a.qsum_partial = h->q_partial;
      RC(head_backward<1>(h, st, a));
    }
    RC(tower_backward(h, st, lc, DQNHIP_CRITIC, nullptr, nullptr, h->act[4], h->dZc, B, false, true, h->S, h->S + kNO, nullptr, fused_seed ? &qr : nullptr));
    // inverting gradients (src/dqn.cpp:924-957) + actor heads backward (src/dqn.cpp:960-963)
    {
      HeadBwdArgs a{}; a.dXc = h->dZc[0]; a.ldx = lc.kp[0]; a.S = h->S; a.aout16 = h->aout16; a.dA16 = h->dA16;
      a.W = wat(h, DQNHIP_ACTOR, la.hw_off); a.X4 = h->act[1][L]; a.H = Hh; a.rows = B; a.dZ = h->dZa[L];
      a.dW = h->g[0] + la.hw_off; a.db = h->g[0] + la.hb_off; a.partial = h->part[0] + la.part_off[L];
      const bool ride = !head_big_ok(h, B, Hh) && head_wgrad_can_ride(la, B);
      HeadWgradRider r{h->dA16, kAP, h->act[1][L], Hh, B, a.dW, a.db, a.partial, Hh / kRiderCW};   // dA16: the post-invert diffs this launch leaves
      if (ride) { a.dW = nullptr; a.db = nullptr; a.partial = nullptr; }
      RC(head_backward<kNO>(h, st, a));
      RC(tower_backward(h, st, la, DQNHIP_ACTOR, h->g[0], h->part[0], h->act[1], h->dZa, B, true, false, 0, -1, ride ? &r : nullptr));
    }
    if (dp) {
      hipLaunchKernelGGL(k_tails, dim3(1), dim3(256), 0, st, (const float*)nullptr, 0,
                         (const double*)h->q_partial, B, inv_batch, (float*)nullptr, actor_tail, (const DevState*)h->st);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  if (phase == 2) {
    // the actor's optimiser pass is the update's last launch: its block 0 also publishes
    // (critic_loss, avg_q) and advances the iteration / sampling counters
    const TickArgs tick{h->st, critic_tail, actor_tail, (const float*)h->loss_partial, h->n_head_blocks,
                        dp ? (const double*)nullptr : (const double*)h->q_partial, B, (float)(B * h->cfg.dp_world), h->stats_dev};
    if (dp) RC(dp_optimiser_step(h, st, 0, actor_tail, &tick));
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

}  // namespace

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
}  // extern "C"
namespace { int dp_destroy_impl(H* h, bool keep_learner); }
extern "C" {
// every captured launch sequence of this learner: the update graphs AND the data-parallel one (it bakes in the Ring
// struct k_gather takes by value and the weight / shared-prefix pointers, exactly as the others do)
static void drop_graphs_fwd(H* h) {
  for (auto& g : h->graph_exec) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  if (h->dp_graph) { hipGraphExecDestroy(h->dp_graph); h->dp_graph = nullptr; }
  if (h->dp_graph_n) { hipGraphExecDestroy(h->dp_graph_n); h->dp_graph_n = nullptr; }
  h->dp_graph_failed = false; h->dp_graph_n_failed = false; h->graph_failed = false;
}

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
  RC(dalloc(&h->Xc_nx, (size_t)B * h->lc.kp[0]));
  h->act[0][0] = h->Xa_n; h->act[1][0] = h->Xa_s; h->act[2][0] = h->Xc_nx; h->act[3][0] = h->Xc_tr; h->act[4][0] = h->Xc_pl;
  for (int p = 0; p < 5; ++p)
    for (int i = 1; i <= L; ++i) RC(dalloc(&h->act[p][i], (size_t)B * layout_of(h, p >= 2).kp[i]));
  for (int i = 0; i <= L; ++i) { RC(dalloc(&h->dZa[i], (size_t)B * h->la.kp[i])); RC(dalloc(&h->dZc[i], (size_t)B * h->lc.kp[i])); }
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
  hipFree(h->Xa_s); hipFree(h->Xa_n); hipFree(h->Xc_tr); hipFree(h->Xc_pl); hipFree(h->Xc_nx);
  for (int p = 0; p < 5; ++p) for (int i = 1; i <= h->L; ++i) hipFree(h->act[p][i]);
  for (int i = 0; i <= h->L; ++i) { hipFree(h->dZa[i]); hipFree(h->dZc[i]); }
  hipFree(h->mb_reward); hipFree(h->mb_mc); hipFree(h->mb_term); hipFree(h->mb_idx);
  hipHostFree(h->idx_pinned); hipHostFree(h->pinned_stats);
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
static int capture_graph(H* h, int which, const int* idx_fixed = nullptr) {
  // Capture phases 0,1,2 once; replays re-read every changing scalar from DevState
  // and (which == 1) the indices from the fixed pinned buffer through a memcpy node; which == 2, 3: the indices
  // are already in the given device buffer (dqnhip_update_pipelined's two slots).
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  const int* idx_dev = idx_fixed;
  if (which == 1) idx_dev = h->idx_pinned_dev;
  const int it_a = h->h_actor_iter, it_c = h->h_critic_iter;
  for (int u = 0; u < (which == 4 ? kMultiU : 1); ++u) {
    h->cap_u = which == 4 ? u : -1;
    for (int p = 0; p < 3 && !rc; ++p) rc = run_phase(h, p, idx_dev);
  }
  h->cap_u = -1;
  h->h_actor_iter = it_a; h->h_critic_iter = it_c;   // capture does not execute
  hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(&h->graph_exec[which], graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) return fail("hipGraphInstantiate: %s", hipGetErrorString(e));
  return 0;
}

int dqnhip_update_async(dqnhip_handle h, const int32_t* idx_host) {
  if (!h) return fail("null handle");
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

// One-deep pipelined form of dqnhip_update: enqueues update t and returns the scalars of update t-1 (zeros on the
// first call).  The host then waits for update t-1 only, while update t is already queued behind it — the device
// never idles on the host's index draw, the H2D of the indices or the read-back, which dqnhip_update pays on
// every call.  Indices and scalars use two pinned slots each (nothing in flight is overwritten).
int dqnhip_update_pipelined(dqnhip_handle h, const int32_t* idx_host, float* critic_loss, float* avg_q) {
  if (!h) return fail("null handle");
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
  auto one = [&]() -> int {
    for (int32_t& i : idx) i = std::uniform_int_distribution<int>(0, size - 1)(rng);
    return pipelined ? dqnhip_update_pipelined(h, idx.data(), &loss, &avgq) : dqnhip_update(h, idx.data(), &loss, &avgq);
  };
  for (int i = 0; i < warmup; ++i) RC(one());
  HIPCHK(hipStreamSynchronize(h->stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iterations; ++i) RC(one());
  HIPCHK(hipStreamSynchronize(h->stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (pipelined) RC(dqnhip_read_stats(h, &loss, &avgq));      // drains the one outstanding read-back
  if (avg_ms) *avg_ms = (float)(ms / iterations);
  return 0;
}

// ---- native data parallelism: RCCL over xGMI inside the library --------------------
// The reference has no collective (threads + one mutex, src/dqn_main.cpp:62-63, 359-363).  Here a
// data-parallel group is one learner per GPU; each rank gathers its own minibatch slice from its
// own replay shard, and the update has exactly two exchange points (the actor step reads the
// UPDATED critic, src/dqn.cpp:904 -> 914): a sum all-reduce of the critic gradient arena after
// phase 0 and of the actor's after phase 1, in place, on the learner's stream — no host sync, no
// Python in the loop.  (q - y)/B uses the global B, the actor gradient is an un-normalised sum
// (src/dqn.cpp:918-921), the clip norm is recomputed on the reduced gradient: every rank applies
// the identical Adam step.  [loss_sum, q_sum] ride in the 4-float arena tails.

#define NCCLCHK(expr)                                                                     \
  do {                                                                                    \
    ncclResult_t r__ = (expr);                                                            \
    if (r__ != ncclSuccess)                                                               \
      return fail("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, __LINE__); \
  } while (0)

}  // extern "C"

namespace {
int dp_broadcast(H* h, int root) {
  for (int net = 0; net < 4; ++net) NCCLCHK(ncclBroadcast(h->w[net], h->w[net], layout_of(h, net).arena, ncclFloat, root, h->comm, h->stream));
  for (int net = 0; net < 2; ++net) {
    NCCLCHK(ncclBroadcast(h->m[net], h->m[net], layout_of(h, net).arena, ncclFloat, root, h->comm, h->stream));
    NCCLCHK(ncclBroadcast(h->v[net], h->v[net], layout_of(h, net).arena, ncclFloat, root, h->comm, h->stream));
  }
  NCCLCHK(ncclBroadcast(&h->st->actor_iter, &h->st->actor_iter, 2, ncclInt32, root, h->comm, h->stream));
  int it[2];
  HIPCHK(hipMemcpyAsync(it, &h->st->actor_iter, sizeof it, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->h_actor_iter = it[0]; h->h_critic_iter = it[1];
  for (int net = 0; net < 4; ++net) h->w16_dirty[net] = true;
  return 0;
}

// per-layer bucket: floats [off, off + count) of net's gradient arena, on the communication stream, ordered after
// everything enqueued on `st` so far
int dp_reduce_slice(H* h, hipStream_t st, int net, size_t off, size_t count) {
  HIPCHK(hipEventRecord(h->comm_ev[0], st));
  HIPCHK(hipStreamWaitEvent(h->comm_stream, h->comm_ev[0], 0));
  float* ptr = h->g[net] + off;
  NCCLCHK(ncclAllReduce(ptr, ptr, count, ncclFloat, ncclSum, h->comm, h->comm_stream));
  return 0;
}

// all-gather (in place) of what the sharded optimiser step of `net` wrote on each rank's slice: the online weights and the
// target's — the targets move on every update (SoftUpdateNet, src/dqn.cpp:967-970), so they cannot be left to an occasional
// broadcast — and, for the fp16 learner, the two fp16 mirrors the GEMMs read.  m and v stay per slice.
int dp_allgather_weights(H* h, int net) {
  size_t lo, hi; shard_range(h, net, lo, hi);
  const size_t n = hi - lo;
  hipStream_t st = h->stream;
  ncclResult_t r = ncclGroupStart();
  if (r == ncclSuccess) r = ncclAllGather(h->w[net] + lo, h->w[net], n, ncclFloat, h->comm, st);
  if (r == ncclSuccess) r = ncclAllGather(h->w[net + 2] + lo, h->w[net + 2], n, ncclFloat, h->comm, st);
  if (h->fp16) {
    if (r == ncclSuccess) r = ncclAllGather(h->w16a[net] + lo, h->w16a[net], n, ncclHalf, h->comm, st);
    if (r == ncclSuccess) r = ncclAllGather(h->w16a[net + 2] + lo, h->w16a[net + 2], n, ncclHalf, h->comm, st);
  }
  const ncclResult_t r2 = ncclGroupEnd();
  if (r != ncclSuccess || r2 != ncclSuccess) return fail("ncclAllGather (sharded optimiser, net %d) failed: %s", net, ncclGetErrorString(r != ncclSuccess ? r : r2));
  return 0;
}

// the exchange step after phase 0 (net = critic) / phase 1 (net = actor)
int dp_exchange(H* h, int net) {
  const NetLayout& l = layout_of(h, net);
  hipStream_t st = h->stream;
  if (h->dp_shard) {
    // reduce-scatter: rank r ends up with floats [r, r + 1) * arena / N of the summed gradient (in place; bf16 on the links
    // under DQNHIP_DP_HALF_GRADS), takes the sum of squares of that slice (the same pass widens a bf16 slice back to fp32),
    // and the ranks all-reduce the 4-float tail {loss, q, target flag, sum of squares}: the clip norm every rank's Adam uses
    size_t lo, hi; shard_range(h, net, lo, hi);
    float* tail = h->dp_tails + (net == DQNHIP_CRITIC ? 0 : 4);
    if (h->dp_half) {
      hipLaunchKernelGGL(k_to_bf16, dim3(1024), dim3(256), 0, st, (const float*)h->g[net], l.arena / 4, h->g16[net]);
      HIPCHK(hipGetLastError());
      NCCLCHK(ncclReduceScatter(h->g16[net], h->g16[net] + lo, hi - lo, ncclBfloat16, ncclSum, h->comm, st));
    } else {
      NCCLCHK(ncclReduceScatter(h->g[net], h->g[net] + lo, hi - lo, ncclFloat, ncclSum, h->comm, st));
    }
    RC(sumsq_launch(h, net, lo, hi));
    hipLaunchKernelGGL(k_shard_scal, dim3(1), dim3(256), 0, st, (const float*)h->part_dp, h->n_part_dp, tail);
    HIPCHK(hipGetLastError());
    NCCLCHK(ncclAllReduce(tail, tail, 4, ncclFloat, ncclSum, h->comm, st));
    return 0;
  }
  if (h->dp_half) {
    // bf16 image of the arena -> sum all-reduce -> (phase 1 / 2 widen it again inside k_sumsq_bf16).  The fp32
    // tails of both nets travel once, with the actor's gradients (nothing reads them before the tick of phase 2).
    hipLaunchKernelGGL(k_to_bf16, dim3(1024), dim3(256), 0, st, (const float*)h->g[net], l.arena / 4, h->g16[net]);
    HIPCHK(hipGetLastError());
    if (net == DQNHIP_CRITIC) {
      NCCLCHK(ncclAllReduce(h->g16[net], h->g16[net], l.arena, ncclBfloat16, ncclSum, h->comm, st));
    } else {
      // one grouped call: the actor's bf16 image and the 8 fp32 tail floats (a failure inside the group still closes it)
      NCCLCHK(ncclGroupStart());
      ncclResult_t r1 = ncclAllReduce(h->g16[net], h->g16[net], l.arena, ncclBfloat16, ncclSum, h->comm, st);
      ncclResult_t r2 = r1 == ncclSuccess ? ncclAllReduce(h->dp_tails, h->dp_tails, 8, ncclFloat, ncclSum, h->comm, st) : r1;
      ncclResult_t r3 = ncclGroupEnd();
      if (r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess)
        return fail("grouped ncclAllReduce (actor gradients + tails) failed: %s", ncclGetErrorString(r1 != ncclSuccess ? r1 : r2 != ncclSuccess ? r2 : r3));
    }
  } else if (h->dp_per_layer) {
    // the tower slices are already in flight on comm_stream; what is left is the head + tail slice, then the main
    // stream waits for the communication stream
    RC(dp_reduce_slice(h, st, net, l.hw_off, l.arena + 4 - l.hw_off));
    HIPCHK(hipEventRecord(h->comm_ev[1], h->comm_stream));
    HIPCHK(hipStreamWaitEvent(st, h->comm_ev[1], 0));
  } else {
    NCCLCHK(ncclAllReduce(h->g[net], h->g[net], l.arena + 4, ncclFloat, ncclSum, h->comm, st));
  }
  return 0;
}

// phase 0, exchange, phase 1, exchange, phase 2 on the learner's stream; no host sync (capturable)
int dp_sequence(H* h, const int* idx_dev) {
  RC(run_phase(h, 0, idx_dev));
  RC(dp_exchange(h, DQNHIP_CRITIC));
  RC(run_phase(h, 1, nullptr));
  RC(dp_exchange(h, DQNHIP_ACTOR));
  return run_phase(h, 2, nullptr);
}

// multi: kMultiU updates in one graph, each gather riding in the previous update's last launch (capture_graph)
int dp_capture(H* h, bool multi = false) {
  hipGraph_t graph = nullptr;
  hipGraphExec_t* out = multi ? &h->dp_graph_n : &h->dp_graph;
  HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  const int it_a = h->h_actor_iter, it_c = h->h_critic_iter;
  int rc = 0;
  for (int u = 0; u < (multi ? kMultiU : 1) && !rc; ++u) {
    h->cap_u = multi ? u : -1;
    rc = dp_sequence(h, nullptr);
  }
  h->cap_u = -1;
  h->h_actor_iter = it_a; h->h_critic_iter = it_c;   // capture does not execute
  hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail("hipStreamEndCapture (dp): %s", hipGetErrorString(e));
  e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) { *out = nullptr; return fail("hipGraphInstantiate (dp): %s", hipGetErrorString(e)); }
  return 0;
}

// ---- file rendezvous (one node, no launcher support) ----------------------------------------------
// Rank r > 0 publishes a request <path>.req<r> holding a fresh random nonce and re-publishes it if it disappears;
// rank 0 first removes whatever an earlier job left behind (<path>, <path>.req*), waits for the world-1 requests,
// and publishes <path> = {id, nonce_1 .. nonce_{world-1}}.  A waiter accepts <path> only if it carries ITS nonce:
// a file left by an earlier job, or written before this waiter existed, can never hand it a dead id.  After the
// group is up (ncclCommInitRank is collective: every rank has read the file by then) rank 0 removes all of it,
// so the same path serves the next group.
struct RvFile { unsigned char id[DQNHIP_DP_ID_BYTES]; uint64_t nonce[64]; };

bool rv_write(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, n, f) == n;
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str())) { unlink(tmp.c_str()); return false; }
  return true;
}
bool rv_read(const std::string& path, void* data, size_t n) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = fread(data, 1, n, f);
  fclose(f);
  return got == n;
}
}  // namespace

extern "C" {

int dqnhip_dp_unique_id(void* id_out, size_t bytes) {
  if (!id_out) return fail("null argument");
  if (bytes != sizeof(ncclUniqueId)) return fail("dp_unique_id: buffer must be DQNHIP_DP_ID_BYTES = %zu bytes", sizeof(ncclUniqueId));
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return 0;
}

int dqnhip_dp_rendezvous_file(const char* path, int32_t rank, int32_t world, int32_t timeout_s, void* id, size_t bytes) {
  if (!path || !id) return fail("null argument");
  if (bytes != DQNHIP_DP_ID_BYTES) return fail("dp_rendezvous_file: id must be DQNHIP_DP_ID_BYTES bytes");
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail("dp_rendezvous_file: bad rank %d / world %d (<= 64)", rank, world);
  const std::string p(path);
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
  auto nap = [] { std::this_thread::sleep_for(std::chrono::milliseconds(10)); };
  if (rank == 0) {
    unlink(p.c_str());
    for (int r = 1; r < world; ++r) unlink((p + ".req" + std::to_string(r)).c_str());
    RvFile f{};
    memcpy(f.id, id, sizeof f.id);
    for (int r = 1; r < world; ++r) {
      const std::string rq = p + ".req" + std::to_string(r);
      while (!rv_read(rq, &f.nonce[r], sizeof(uint64_t)) || f.nonce[r] == 0) {
        if (expired()) return fail("dp_rendezvous_file: rank 0 timed out after %d s waiting for rank %d (%s)", timeout_s, r, rq.c_str());
        nap();
      }
    }
    if (!rv_write(p, &f, sizeof f)) return fail("dp_rendezvous_file: cannot publish %s", path);
    return 0;
  }
  std::random_device rd;
  uint64_t nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 17) ^
                   (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  if (nonce == 0) nonce = 1;
  const std::string rq = p + ".req" + std::to_string(rank);
  for (;;) {
    uint64_t seen = 0;
    if (!rv_read(rq, &seen, sizeof seen) || seen != nonce) {            // not there (yet, or rank 0 cleaned up): (re)publish
      if (!rv_write(rq, &nonce, sizeof nonce)) return fail("dp_rendezvous_file: cannot write %s", rq.c_str());
    }
    RvFile f{};
    if (rv_read(p, &f, sizeof f) && f.nonce[rank] == nonce) { memcpy(id, f.id, sizeof f.id); return 0; }
    if (expired()) return fail("dp_rendezvous_file: rank %d timed out after %d s waiting for %s", rank, timeout_s, path);
    nap();
  }
}

int dqnhip_dp_rendezvous_cleanup(const char* path, int32_t world) {
  if (!path) return fail("null argument");
  const std::string p(path);
  unlink(p.c_str());
  for (int r = 1; r < world; ++r) unlink((p + ".req" + std::to_string(r)).c_str());
  return 0;
}

int dqnhip_dp_init(dqnhip_handle h, const void* id, size_t bytes, int32_t flags) {
  if (!h || !id) return fail("null argument");
  if (bytes != sizeof(ncclUniqueId)) return fail("dp_init: id must be DQNHIP_DP_ID_BYTES = %zu bytes", sizeof(ncclUniqueId));
  if (h->comm) return fail("dp_init: this learner already has a communicator");
  if (h->w_owner || h->sharers) return fail("dp_init: learners that share layers cannot join a data-parallel group");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
  NCCLCHK(ncclCommInitRank(&h->comm, h->cfg.dp_world, uid, h->cfg.dp_rank));
  h->dp_half = (flags & DQNHIP_DP_HALF_GRADS) != 0;
  h->dp_shard = (flags & DQNHIP_DP_SHARD_OPT) != 0;
  if (h->dp_shard)
    for (int net = 0; net < 2; ++net)
      if (layout_of(h, net).arena % ((size_t)4 * h->cfg.dp_world)) {
        ncclCommDestroy(h->comm); h->comm = nullptr; h->dp_half = h->dp_shard = false;
        return fail("dp_init: DQNHIP_DP_SHARD_OPT needs a parameter arena (%zu floats) divisible by 4 x dp_world = %d", layout_of(h, net).arena, 4 * h->cfg.dp_world);
      }
  // per-layer buckets need each layer's dW AND db final when its backward launch has run: true for the fp32 path;
  // the fp16 path produces all wgrads of a net in one launch at the end and keeps one collective per net
  h->dp_per_layer = (flags & DQNHIP_DP_PER_LAYER) != 0 && !h->fp16 && !h->dp_half && !h->dp_shard;
  HIPCHK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  for (auto& e : h->comm_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (h->dp_half)
    for (int net = 0; net < 2; ++net) HIPCHK(hipMalloc(&h->g16[net], layout_of(h, net).arena * sizeof(uint16_t)));
  if (h->dp_half || h->dp_shard) {
    HIPCHK(hipMalloc(&h->dp_tails, 8 * sizeof(float)));
    HIPCHK(hipMemsetAsync(h->dp_tails, 0, 8 * sizeof(float), h->stream));
  }
  drop_graphs_fwd(h);
  // replicas start from rank 0's state: weights of the four nets, Adam history, iterations
  return dp_broadcast(h, 0);
}

// Single-node rendezvous without any launcher support (dqnhip_dp_rendezvous_file), then dqnhip_dp_init.
// (A launcher that has its own channel — MPI, torch.distributed's store — passes the id to dqnhip_dp_init directly.)
int dqnhip_dp_init_file(dqnhip_handle h, const char* path, int32_t flags, int32_t timeout_s) {
  if (!h || !path) return fail("null argument");
  ncclUniqueId uid;
  if (h->cfg.dp_rank == 0) RC(dqnhip_dp_unique_id(&uid, sizeof uid));
  RC(dqnhip_dp_rendezvous_file(path, h->cfg.dp_rank, h->cfg.dp_world, timeout_s, &uid, sizeof uid));
  const int rc = dqnhip_dp_init(h, &uid, sizeof uid, flags);
  // ncclCommInitRank is collective: once it has returned on rank 0 every rank has read the file
  if (h->cfg.dp_rank == 0) { const std::string msg = g_err; dqnhip_dp_rendezvous_cleanup(path, h->cfg.dp_world); g_err = msg; }
  return rc;
}

int dqnhip_dp_broadcast_params(dqnhip_handle h, int32_t root) {
  if (!h) return fail("null handle");
  if (!h->comm) return fail("dp_broadcast_params: no communicator (call dqnhip_dp_init first)");
  if (root < 0 || root >= h->cfg.dp_world) return fail("bad root %d", root);
  HIPCHK(hipSetDevice(h->cfg.device));
  return dp_broadcast(h, root);
}

int dqnhip_dp_update(dqnhip_handle h, const int32_t* idx_host) {
  if (!h) return fail("null handle");
  if (!h->comm) return fail("dp_update: no communicator (call dqnhip_dp_init first)");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->next_phase != 0) return fail("dqnhip_dp_update: a phased update is in progress (next phase %d)", h->next_phase);
  RingUse ring_use(h);
  RC(sync_dirty16(h));
  if (h->dp_shard) h->shard_stale = true;
  // cfg.use_graph: the whole update — 30-40 launches and both collectives — replays as ONE hipGraph (every rank
  // captures the same sequence).  Explicit indices, kernel timing, or a capture that RCCL refuses: eager.
  if (h->cfg.use_graph && !idx_host && !h->timing && !h->dp_graph_failed) {
    if (RO(h)->h_size < 1) RC(refresh_ring(h));
    if (RO(h)->h_size < 1) return fail("replay memory is empty");
    if (!h->dp_graph && dp_capture(h)) h->dp_graph_failed = true;
    if (h->dp_graph) {
      HIPCHK(hipGraphLaunch(h->dp_graph, h->stream));
      h->h_actor_iter += 1; h->h_critic_iter += 1;
      return 0;
    }
  }
  const int* idx_dev = nullptr;
  RC(stage_indices(h, idx_host, &idx_dev));
  return dp_sequence(h, idx_dev);
}

// n data-parallel updates with on-device sampling (dqnhip_update_async_n for a group: every rank calls it with the same n)
int dqnhip_dp_update_n(dqnhip_handle h, int32_t n) {
  if (!h) return fail("null handle");
  if (!h->comm) return fail("dp_update_n: no communicator (call dqnhip_dp_init first)");
  if (n < 0) return fail("dqnhip_dp_update_n: n must be >= 0");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->next_phase != 0) return fail("dqnhip_dp_update_n: a phased update is in progress (next phase %d)", h->next_phase);
  if (h->cfg.use_graph && !h->timing && !h->dp_graph_failed && !h->dp_graph_n_failed && n >= kMultiU) {
    RingUse ring_use(h);
    RC(sync_dirty16(h));
    if (RO(h)->h_size < 1) RC(refresh_ring(h));
    if (RO(h)->h_size < 1) return fail("replay memory is empty");
    if (!h->dp_graph_n && dp_capture(h, true)) h->dp_graph_n_failed = true;
    while (n >= kMultiU && h->dp_graph_n) {
      if (h->dp_shard) h->shard_stale = true;
      HIPCHK(hipGraphLaunch(h->dp_graph_n, h->stream));
      h->h_actor_iter += kMultiU; h->h_critic_iter += kMultiU; n -= kMultiU;
    }
  }
  for (; n > 0; --n) RC(dqnhip_dp_update(h, nullptr));
  return 0;
}

int dqnhip_dp_graph_active(dqnhip_handle h, int32_t* active) {
  if (!h || !active) return fail("null argument");
  *active = h->dp_graph != nullptr;
  return 0;
}

// Sharded optimiser: every rank holds m and v of its own slice only; this all-gathers them (collective: every rank of the
// group calls it) so that dqnhip_get_params(KIND_M / KIND_V), a snapshot, or a later replicated / single-learner update
// see the whole Adam history.  No-op without DQNHIP_DP_SHARD_OPT.
int dqnhip_dp_gather_state(dqnhip_handle h) {
  if (!h) return fail("null handle");
  if (!h->comm) return fail("dp_gather_state: no communicator (call dqnhip_dp_init first)");
  if (!h->dp_shard) return 0;
  HIPCHK(hipSetDevice(h->cfg.device));
  for (int net = 0; net < 2; ++net) {
    size_t lo, hi; shard_range(h, net, lo, hi);
    NCCLCHK(ncclAllGather(h->m[net] + lo, h->m[net], hi - lo, ncclFloat, h->comm, h->stream));
    NCCLCHK(ncclAllGather(h->v[net] + lo, h->v[net], hi - lo, ncclFloat, h->comm, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->shard_stale = false;
  return 0;
}

}  // extern "C"
namespace {
// keep_learner: the learner lives on as a plain one -> its Adam history must be whole.  The gather that makes it whole is a
// COLLECTIVE, and a teardown must never block on peers that may be gone: it is asked for, not done implicitly.
int dp_destroy_impl(H* h, bool keep_learner) {
  if (!h || !h->comm) return 0;
  if (keep_learner && h->dp_shard && h->shard_stale)
    return fail("dqnhip_dp_destroy: the optimiser is sharded and updates ran since the last dqnhip_dp_gather_state — call it on every rank first "
                "(this rank holds the Adam history of its own slice only)");
  hipSetDevice(h->cfg.device);
  hipStreamSynchronize(h->stream);
  hipStreamSynchronize(h->comm_stream);
  if (h->dp_graph) { hipGraphExecDestroy(h->dp_graph); h->dp_graph = nullptr; }
  if (h->dp_graph_n) { hipGraphExecDestroy(h->dp_graph_n); h->dp_graph_n = nullptr; }
  h->dp_graph_failed = false; h->dp_graph_n_failed = false;
  ncclCommDestroy(h->comm); h->comm = nullptr;
  hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr;
  for (auto& e : h->comm_ev) { if (e) hipEventDestroy(e); e = nullptr; }
  for (int net = 0; net < 2; ++net) if (h->g16[net]) { hipFree(h->g16[net]); h->g16[net] = nullptr; }
  if (h->dp_tails) { hipFree(h->dp_tails); h->dp_tails = nullptr; }
  h->dp_half = false; h->dp_per_layer = false; h->dp_shard = false; h->shard_stale = false;
  return 0;
}
}  // namespace
extern "C" {
int dqnhip_dp_destroy(dqnhip_handle h) { return dp_destroy_impl(h, true); }

// ---- acting ------------------------------------------------------------------------

static int actor_forward_dev(H* h, int net, const float* states_dev, int n, float* out_dev) {
  if (n < 1) return fail("n must be >= 1");
  if (net != DQNHIP_ACTOR && net != DQNHIP_ACTOR_TARGET) return fail("net must be an actor");
  const int rows = round_up(n, 32);
  RC(ensure_act(h, rows));
  const NetLayout& l = h->la;
  float* acts[kMaxL + 1];
  float* p = h->act_buf;
  for (int i = 0; i <= l.L; ++i) { acts[i] = p; p += (size_t)rows * std::max(h->la.kp[i], h->lc.kp[i]); }
  float* out16 = p;
  hipLaunchKernelGGL(k_pack_rows, dim3((rows * l.kp[0] + 255) / 256), dim3(256), 0, h->stream, states_dev, n,
                     h->S, acts[0], rows, l.kp[0]);
  HIPCHK(hipGetLastError());
  FwdPass fp[1] = {{net, &l, acts}};
  RC(tower_forward(h, h->stream, fp, 1, rows));
  HeadArgs a{}; a.X = acts[l.L]; a.ldx = l.dims[l.L]; a.H = l.dims[l.L]; a.rows = rows;
  a.W = wat(h, net, l.hw_off); a.b = wat(h, net, l.hb_off); a.out16 = out16;
  RC((head_forward<kNO, HEAD_ACTOR>(h, h->stream, a)));
  hipLaunchKernelGGL(k_unpack_out, dim3((n * kNO + 255) / 256), dim3(256), 0, h->stream, (const float*)out16, n, out_dev);
  HIPCHK(hipGetLastError());
  return 0;
}

int dqnhip_select_actions_device(dqnhip_handle h, const float* states_dev, int32_t n, float* actor_out_dev) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  return actor_forward_dev(h, DQNHIP_ACTOR, states_dev, n, actor_out_dev);
}

int dqnhip_select_actions_net(dqnhip_handle h, int32_t net, const float* states_host, int32_t n, float* actor_out_host) {
  if (!h) return fail("null handle");
  if (n < 1) return fail("n must be >= 1");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t sb = (size_t)n * h->S * sizeof(float), ob = (size_t)n * kNO * sizeof(float);
  RC(ensure_stage(h, round_up_z(sb, 256) + ob));
  float* sdev = (float*)h->stage_dev;
  float* odev = (float*)((char*)h->stage_dev + round_up_z(sb, 256));
  HIPCHK(hipMemcpyAsync(sdev, states_host, sb, hipMemcpyHostToDevice, h->stream));
  RC(actor_forward_dev(h, net, sdev, n, odev));
  HIPCHK(hipMemcpyAsync(actor_out_host, odev, ob, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int dqnhip_select_actions(dqnhip_handle h, const float* states_host, int32_t n, float* actor_out_host) {
  return dqnhip_select_actions_net(h, DQNHIP_ACTOR, states_host, n, actor_out_host);
}

int dqnhip_critic_forward(dqnhip_handle h, int32_t net, const float* states_host, const float* actor_out_host,
                          int32_t n, float* q_host) {
  if (!h) return fail("null handle");
  if (n < 1) return fail("n must be >= 1");
  if (net != DQNHIP_CRITIC && net != DQNHIP_CRITIC_TARGET) return fail("net must be a critic");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int rows = round_up(n, 32);
  const size_t sb = round_up_z((size_t)n * h->S * sizeof(float), 256), ab = round_up_z((size_t)n * kNO * sizeof(float), 256);
  RC(ensure_stage(h, sb + ab + (size_t)rows * sizeof(float)));
  float* sdev = (float*)h->stage_dev;
  float* adev = (float*)((char*)h->stage_dev + sb);
  float* qdev = (float*)((char*)h->stage_dev + sb + ab);
  HIPCHK(hipMemcpyAsync(sdev, states_host, (size_t)n * h->S * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(adev, actor_out_host, (size_t)n * kNO * sizeof(float), hipMemcpyHostToDevice, h->stream));
  RC(ensure_act(h, rows));
  const NetLayout& l = h->lc;
  float* acts[kMaxL + 1];
  float* p = h->act_buf;
  for (int i = 0; i <= l.L; ++i) { acts[i] = p; p += (size_t)rows * std::max(h->la.kp[i], h->lc.kp[i]); }
  hipLaunchKernelGGL(k_pack_critic, dim3((rows * l.kp[0] + 255) / 256), dim3(256), 0, h->stream, (const float*)sdev,
                     (const float*)adev, n, h->S, acts[0], rows, l.kp[0]);
  HIPCHK(hipGetLastError());
  FwdPass fp[1] = {{net, &l, acts}};
  RC(tower_forward(h, h->stream, fp, 1, rows));
  HeadArgs a{}; a.X = acts[l.L]; a.ldx = l.dims[l.L]; a.H = l.dims[l.L]; a.rows = rows;
  a.W = wat(h, net, l.hw_off); a.b = wat(h, net, l.hb_off); a.q = qdev;
  RC((head_forward<1, HEAD_Q>(h, h->stream, a)));
  HIPCHK(hipMemcpyAsync(q_host, qdev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- replay memory -------------------------------------------------------------------

static int add_dev(H* h, const float* s, const float* a, const float* r, const float* mc, const float* nx,
                   const uint8_t* term, int n, int single) {
  if (n < 1) return fail("n must be >= 1");
  RingUse ring_use(h);
  RC(refresh_ring(h));
  const long long cap = RO(h)->ring.cap;
  if (single == 0 && n >= cap) return fail("AddTransitions: batch of %d does not fit capacity %lld (the reference would pop an empty deque)", n, cap);
  if (single == 2 && RO(h)->h_size + n > cap) return fail("LoadReplayMemory: %lld transitions exceed the capacity %lld", RO(h)->h_size + n, cap);
  hipLaunchKernelGGL(k_add_transitions, dim3((n + 3) / 4), dim3(256), 0, h->stream, RO(h)->ring, RO(h)->st, s, a, r, mc, nx,
                     term, n, single, RO(h)->done_counter);
  HIPCHK(hipGetLastError());
  // host mirror of the same deque arithmetic (src/dqn.cpp:768-781)
  if (single == 2) { }
  else if (single) { if (RO(h)->h_size == cap) { RO(h)->h_head = (RO(h)->h_head + 1) % cap; RO(h)->h_size -= 1; } }
  else {
    long long pops = RO(h)->h_size + n - cap + 1;
    pops = std::max(0LL, std::min(pops, RO(h)->h_size));
    RO(h)->h_head = (RO(h)->h_head + pops) % cap; RO(h)->h_size -= pops;
  }
  RO(h)->h_size += n;
  return 0;
}

int dqnhip_add_transitions_device(dqnhip_handle h, const float* states, const float* actor_out, const float* rewards,
                                  const float* on_policy_targets, const float* next_states, const uint8_t* terminal,
                                  int32_t n) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  return add_dev(h, states, actor_out, rewards, on_policy_targets, next_states, terminal, n, 0);
}

static int add_host(H* h, const float* s, const float* a, const float* r, const float* mc, const float* nx,
                    const uint8_t* term, int n, int single) {
  if (!h) return fail("null handle");
  if (n < 1) return fail("n must be >= 1");
  if (!s || !a || !r || !mc || !term) return fail("null input array");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t sb = round_up_z((size_t)n * h->S * 4, 256), ab = round_up_z((size_t)n * kNO * 4, 256), vb = round_up_z((size_t)n * 4, 256);
  // staging is reused: wait for the previous scatter to drain before overwriting
  HIPCHK(hipStreamSynchronize(h->stream));
  RC(ensure_stage(h, 2 * sb + ab + 3 * vb));
  char* base = (char*)h->stage_dev;
  float* ds = (float*)base; float* dn = (float*)(base + sb); float* da = (float*)(base + 2 * sb);
  float* dr = (float*)(base + 2 * sb + ab); float* dm = (float*)(base + 2 * sb + ab + vb);
  uint8_t* dt = (uint8_t*)(base + 2 * sb + ab + 2 * vb);
  HIPCHK(hipMemcpyAsync(ds, s, (size_t)n * h->S * 4, hipMemcpyHostToDevice, h->stream));
  if (nx) HIPCHK(hipMemcpyAsync(dn, nx, (size_t)n * h->S * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(da, a, (size_t)n * kNO * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dr, r, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dm, mc, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dt, term, (size_t)n, hipMemcpyHostToDevice, h->stream));
  return add_dev(h, ds, da, dr, dm, nx ? dn : nullptr, dt, n, single);
}

int dqnhip_add_transitions(dqnhip_handle h, const float* states, const float* actor_out, const float* rewards,
                           const float* on_policy_targets, const float* next_states, const uint8_t* terminal, int32_t n) {
  return add_host(h, states, actor_out, rewards, on_policy_targets, next_states, terminal, n, 0);
}

int dqnhip_add_transition(dqnhip_handle h, const float* state, const float* actor_out, float reward,
                          float on_policy_target, const float* next_state, uint8_t terminal) {
  return add_host(h, state, actor_out, &reward, &on_policy_target, next_state, &terminal, 1, 1);
}

int dqnhip_label_transitions(double gamma, const float* rewards, int32_t n, float* mc) {
  if (n < 1) return fail("Need at least one transition to label.");   // CHECK_GT, src/dqn.cpp:784
  if (!rewards || !mc) return fail("null array");
  mc[n - 1] = rewards[n - 1];
  for (int i = n - 2; i >= 0; --i) mc[i] = (float)((double)rewards[i] + gamma * (double)mc[i + 1]);
  return 0;
}

int dqnhip_memory_size(dqnhip_handle h, int32_t* size) {
  if (!h || !size) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  RingUse ring_use(h);
  RC(refresh_ring(h));
  *size = (int32_t)RO(h)->h_size;
  return 0;
}

int dqnhip_clear_memory(dqnhip_handle h) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  RingUse ring_use(h);
  HIPCHK(hipMemsetAsync(RO(h)->st, 0, 2 * sizeof(int), h->stream));   // ring_head, ring_size
  RO(h)->h_head = 0; RO(h)->h_size = 0; RO(h)->ring_stale = false;
  return 0;
}

static bool same_nets(const dqnhip_learner* a, const dqnhip_learner* b);

// caller holds the RingUse of h and has refreshed (head,size)
static int read_memory_impl(H* h, int32_t first, int32_t n, float* states, float* actor_out, float* rewards,
                            float* on_policy_targets, float* next_states, uint8_t* terminal) {
  if (n < 1 || first < 0 || (long long)first + n > RO(h)->h_size) return fail("read_memory range [%d,%d) outside [0,%lld)", first, first + n, RO(h)->h_size);
  const size_t sb = round_up_z((size_t)n * h->S * 4, 256), ab = round_up_z((size_t)n * kNO * 4, 256), vb = round_up_z((size_t)n * 4, 256);
  HIPCHK(hipStreamSynchronize(h->stream));
  RC(ensure_stage(h, 2 * sb + ab + 3 * vb));
  char* base = (char*)h->stage_dev;
  float* ds = (float*)base; float* dn = (float*)(base + sb); float* da = (float*)(base + 2 * sb);
  float* dr = (float*)(base + 2 * sb + ab); float* dm = (float*)(base + 2 * sb + ab + vb);
  uint8_t* dt = (uint8_t*)(base + 2 * sb + ab + 2 * vb);
  hipLaunchKernelGGL(k_read_memory, dim3((n + 3) / 4), dim3(256), 0, h->stream, RO(h)->ring, (const DevState*)RO(h)->st, first, n,
                     ds, da, dr, dm, dn, dt);
  HIPCHK(hipGetLastError());
  if (states) HIPCHK(hipMemcpyAsync(states, ds, (size_t)n * h->S * 4, hipMemcpyDeviceToHost, h->stream));
  if (next_states) HIPCHK(hipMemcpyAsync(next_states, dn, (size_t)n * h->S * 4, hipMemcpyDeviceToHost, h->stream));
  if (actor_out) HIPCHK(hipMemcpyAsync(actor_out, da, (size_t)n * kNO * 4, hipMemcpyDeviceToHost, h->stream));
  if (rewards) HIPCHK(hipMemcpyAsync(rewards, dr, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  if (on_policy_targets) HIPCHK(hipMemcpyAsync(on_policy_targets, dm, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
  if (terminal) HIPCHK(hipMemcpyAsync(terminal, dt, (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int dqnhip_read_memory(dqnhip_handle h, int32_t first, int32_t n, float* states, float* actor_out, float* rewards,
                       float* on_policy_targets, float* next_states, uint8_t* terminal) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  RingUse ring_use(h);
  RC(refresh_ring(h));
  return read_memory_impl(h, first, n, states, actor_out, rewards, on_policy_targets, next_states, terminal);
}

// DQN::SampleStatesFromMemory (src/dqn.cpp:511-523): n states of uniformly sampled transitions.
// idx_host = the explicit form of SampleTransitionsFromMemory (as in dqnhip_update); NULL draws on
// the device from the counter-based generator (its own key stream, one counter tick per call).
int dqnhip_sample_states(dqnhip_handle h, const int32_t* idx_host, int32_t n, float* states_host) {
  if (!h || !states_host) return fail("null argument");
  if (n < 1) return fail("n must be >= 1");
  HIPCHK(hipSetDevice(h->cfg.device));
  RingUse ring_use(h);
  RC(refresh_ring(h));
  const long long size = RO(h)->h_size;
  if (size < 1) return fail("replay memory is empty");
  const size_t ib = round_up_z((size_t)n * sizeof(int), 256), sb = (size_t)n * h->S * sizeof(float);
  HIPCHK(hipStreamSynchronize(h->stream));
  RC(ensure_stage(h, ib + sb));
  int* di = (int*)h->stage_dev; float* ds = (float*)((char*)h->stage_dev + ib);
  if (idx_host) {
    for (int i = 0; i < n; ++i)
      if (idx_host[i] < 0 || idx_host[i] >= size) return fail("sampled index %d = %d out of range [0,%lld)", i, idx_host[i], size);
    HIPCHK(hipMemcpyAsync(di, idx_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  hipLaunchKernelGGL(k_sample_states, dim3((n + 3) / 4), dim3(256), 0, h->stream, RO(h)->ring, (const DevState*)RO(h)->st,
                     idx_host ? (const int*)di : (const int*)nullptr, sample_key(h) ^ 0x5354415445535F5Full, h->sample_states_calls, n, ds);
  HIPCHK(hipGetLastError());
  if (!idx_host) h->sample_states_calls += 1;
  HIPCHK(hipMemcpyAsync(states_host, ds, sb, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// getActorOutput (src/dqn.cpp:719-732): the first `batch_size` rows of an actor's output blobs as
// left by its last forward — here the last update's minibatch forward (ACTOR: mu(s), ACTOR_TARGET: mu'(s')).
int dqnhip_get_actor_output(dqnhip_handle h, int32_t net, int32_t batch_size, float* actor_out_host) {
  if (!h || !actor_out_host) return fail("null argument");
  if (net != DQNHIP_ACTOR && net != DQNHIP_ACTOR_TARGET) return fail("net must be an actor");
  if (batch_size < 1 || batch_size > h->B) return fail("batch_size %d outside [1, %d]", batch_size, h->B);
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<float> tmp((size_t)batch_size * kAP);
  HIPCHK(hipMemcpyAsync(tmp.data(), net == DQNHIP_ACTOR ? h->aout16 : h->aout_t16, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int r = 0; r < batch_size; ++r) memcpy(actor_out_host + (size_t)r * kNO, &tmp[(size_t)r * kAP], kNO * sizeof(float));
  return 0;
}

// Sum the gradient arenas (4-float tails included) of n co-located learners of one data-parallel
// group in rank order and leave the sum in every one of them: the exchange step of
// dqnhip_update_phase for learners that share a device (multi-agent layouts, and the one-GPU parity
// test of the dp_world > 1 code path).  Cross-device groups use dqnhip_dp_* (RCCL).
int dqnhip_reduce_gradients_local(dqnhip_handle* hs, int32_t n, int32_t net) {
  if (!hs || n < 1 || n > 8) return fail("reduce_gradients_local: 1..8 learners");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  LocalReduce a{}; a.n = n;
  for (int i = 0; i < n; ++i) {
    if (!hs[i]) return fail("null handle");
    if (hs[i]->cfg.device != hs[0]->cfg.device || !same_nets(hs[i], hs[0])) return fail("reduce_gradients_local: learners must share a device and a shape");
    a.g[i] = hs[i]->g[net];
  }
  a.n4 = (layout_of(hs[0], net).arena + 4) / 4;
  HIPCHK(hipSetDevice(hs[0]->cfg.device));
  for (int i = 1; i < n; ++i) HIPCHK(hipStreamSynchronize(hs[i]->stream));   // their phase must be complete
  hipLaunchKernelGGL(k_local_reduce, dim3(1024), dim3(256), 0, hs[0]->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(hs[0]->stream));
  return 0;
}

// ---- .replaymemory files (src/dqn.cpp:1146-1226) --------------------------------------------
int dqnhip_snapshot_replay_memory(dqnhip_handle h, const char* filename) {
  if (!h || !filename) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  // one RingUse for the whole file: with a shared ring another agent's AddTransitions must not
  // move the head between chunks (the file would hold shifted / duplicated transitions)
  RingUse ring_use(h);
  RC(refresh_ring(h));
  gzFile f = gzopen(filename, "wb");
  if (!f) return fail("cannot open %s for writing", filename);
  const int32_t n = (int32_t)RO(h)->h_size;
  const size_t S = h->S;
  bool ok = gzwrite(f, &n, sizeof n) == (int)sizeof n;
  const int chunk = 65536;
  std::vector<float> s((size_t)chunk * S), a((size_t)chunk * kNO), r(chunk), mc(chunk);
  std::vector<uint8_t> term(chunk), rec;
  for (int first = 0; first < n && ok; first += chunk) {
    const int m = std::min(chunk, n - first);
    if (read_memory_impl(h, first, m, s.data(), a.data(), r.data(), mc.data(), nullptr, term.data())) { gzclose(f); return 1; }
    const size_t rb = S * 4 + kNO * 4 + 4 + 4 + 1;
    rec.resize((size_t)m * rb);
    for (int i = 0; i < m; ++i) {
      uint8_t* p = &rec[(size_t)i * rb];
      memcpy(p, &s[(size_t)i * S], S * 4); p += S * 4;
      memcpy(p, &a[(size_t)i * kNO], kNO * 4); p += kNO * 4;     // sizeof(ActorOutput)
      memcpy(p, &r[i], 4); p += 4;
      memcpy(p, &mc[i], 4); p += 4;
      *p = term[i] ? 1 : 0;                                       // sizeof(bool) == 1
    }
    ok = gzwrite(f, rec.data(), (unsigned)rec.size()) == (int)rec.size();
  }
  if (gzclose(f) != Z_OK || !ok) return fail("short write to %s", filename);
  return 0;
}

int dqnhip_load_replay_memory(dqnhip_handle h, const char* filename) {
  if (!h || !filename) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  gzFile f = gzopen(filename, "rb");
  if (!f) return fail("Invalid file: %s", filename);              // CHECK(is_regular_file), src/dqn.cpp:1181
  int32_t n = 0;
  if (gzread(f, &n, sizeof n) != (int)sizeof n || n < 0) { gzclose(f); return fail("%s: bad header", filename); }
  if (n > RO(h)->ring.cap) { gzclose(f); return fail("%s holds %d transitions, capacity is %d", filename, n, RO(h)->ring.cap); }
  RC(dqnhip_clear_memory(h));
  const size_t S = h->S, rb = S * 4 + kNO * 4 + 4 + 4 + 1;
  const int chunk = 65536;
  // one record of look-ahead: next state of the last row of a chunk is the first state of the next
  std::vector<uint8_t> rec((size_t)(chunk + 1) * rb);
  std::vector<float> s((size_t)chunk * S), nx((size_t)chunk * S), a((size_t)chunk * kNO), r(chunk), mc(chunk);
  std::vector<uint8_t> term(chunk);
  int have = 0;                       // records buffered in rec
  int done = 0;
  while (done < n) {
    const int want = std::min(chunk + 1, n - done) - have;
    if (want > 0) {
      const int got = gzread(f, &rec[(size_t)have * rb], (unsigned)((size_t)want * rb));
      if (got != (int)((size_t)want * rb)) { gzclose(f); return fail("%s: truncated", filename); }
      have += want;
    }
    const int m = std::min(chunk, n - done);
    for (int i = 0; i < m; ++i) {
      const uint8_t* p = &rec[(size_t)i * rb];
      memcpy(&s[(size_t)i * S], p, S * 4); p += S * 4;
      memcpy(&a[(size_t)i * kNO], p, kNO * 4); p += kNO * 4;
      memcpy(&r[i], p, 4); p += 4;
      memcpy(&mc[i], p, 4); p += 4;
      bool t = *p != 0;
      const bool has_next = done + i + 1 < n;
      if (!t && !has_next) t = true;                              // trailing non-terminal: next stays none
      term[i] = t ? 1 : 0;
      if (!t) memcpy(&nx[(size_t)i * S], &rec[(size_t)(i + 1) * rb], S * 4);
      else memset(&nx[(size_t)i * S], 0, S * 4);
    }
    if (add_host(h, s.data(), a.data(), r.data(), mc.data(), nx.data(), term.data(), m, 2)) { gzclose(f); return 1; }
    // keep the look-ahead record as the first record of the next chunk
    if (have > m) memmove(&rec[0], &rec[(size_t)m * rb], rb);
    have -= m;
    done += m;
  }
  gzclose(f);
  return 0;
}

// ---- parameters ----------------------------------------------------------------------

static float* arena_ptr(H* h, int net, int kind) {
  if (kind == DQNHIP_KIND_W) return (net >= 0 && net < 4) ? h->w[net] : nullptr;
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return nullptr;
  return kind == DQNHIP_KIND_M ? h->m[net] : kind == DQNHIP_KIND_V ? h->v[net] : kind == DQNHIP_KIND_G ? h->g[net] : nullptr;
}

int dqnhip_param_count(dqnhip_handle h, int32_t net, size_t* count) {
  if (!h || !count) return fail("null argument");
  if (net < 0 || net > 3) return fail("bad net %d", net);
  *count = layout_of(h, net).dense;
  return 0;
}

int dqnhip_get_params(dqnhip_handle h, int32_t net, int32_t kind, float* host, size_t count) {
  if (!h || !host) return fail("null argument");
  float* p = arena_ptr(h, net, kind);
  if (!p) return fail("bad (net,kind) = (%d,%d)", net, kind);
  const NetLayout& l = layout_of(h, net);
  if (count != l.dense) return fail("count %zu != parameter count %zu", count, l.dense);
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<float> arena(l.arena);
  const size_t sh = kind == DQNHIP_KIND_W ? h->shared_fl[net & 1] : 0;   // shared first layers: the owner's storage
  if (sh) HIPCHK(hipMemcpyAsync(arena.data(), h->w_owner->w[net], sh * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (sh < l.arena) HIPCHK(hipMemcpyAsync(arena.data() + sh, p + sh, (l.arena - sh) * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  arena_to_dense(l, arena, host);
  return 0;
}

int dqnhip_set_params(dqnhip_handle h, int32_t net, int32_t kind, const float* host, size_t count) {
  if (!h || !host) return fail("null argument");
  float* p = arena_ptr(h, net, kind);
  if (!p) return fail("bad (net,kind) = (%d,%d)", net, kind);
  const NetLayout& l = layout_of(h, net);
  if (count != l.dense) return fail("count %zu != parameter count %zu", count, l.dense);
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<float> arena;
  dense_to_arena(l, host, arena);
  const size_t sh = kind == DQNHIP_KIND_W ? h->shared_fl[net & 1] : 0;
  if (sh) HIPCHK(hipMemcpyAsync(h->w_owner->w[net], arena.data(), sh * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (sh < l.arena) HIPCHK(hipMemcpyAsync(p + sh, arena.data() + sh, (l.arena - sh) * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (kind == DQNHIP_KIND_W) h->w16_dirty[net] = true;
  return 0;
}

int dqnhip_clone_to_target(dqnhip_handle h, int32_t net) {
  if (!h) return fail("null handle");
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t sh = h->shared_fl[net], n = layout_of(h, net).arena;
  if (sh) HIPCHK(hipMemcpyAsync(h->w_owner->w[net + 2], h->w_owner->w[net], sh * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  if (sh < n) HIPCHK(hipMemcpyAsync(h->w[net + 2] + sh, h->w[net] + sh, (n - sh) * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  h->w16_dirty[net + 2] = true;
  return 0;
}

// ---- multi-agent sharing (src/dqn.cpp:1036-1083, src/dqn_main.cpp:305-323) ------------------

static void drop_graphs(H* h) { drop_graphs_fwd(h); }

static bool same_nets(const H* a, const H* b) {
  if (a->S != b->S || a->L != b->L) return false;
  for (int i = 0; i < a->L; ++i) if (a->cfg.hidden[i] != b->cfg.hidden[i]) return false;
  return true;
}

// floats of the arena covered by the first `n` layers-with-blobs of a net (Caffe layer order:
// ip1..ipL, then action_layer, actionpara_layer / q_values_layer)
static int shared_prefix(const NetLayout& l, int n, size_t* fl) {
  const int heads = l.NH == kNO ? 2 : 1;
  if (n < 0 || n > l.L + heads) return fail("cannot share %d layers of a net with %d", n, l.L + heads);   // CHECK_LT, src/dqn.cpp:1060
  if (n < l.L) *fl = l.w_off[n];
  else if (n == l.L) *fl = l.hw_off;
  else if (n == l.L + heads) *fl = l.arena;
  else return fail("sharing action_layer without actionpara_layer is not supported (the two heads are one [10][H] matrix here)");
  return 0;
}

int dqnhip_share_parameters(dqnhip_handle owner, dqnhip_handle other, int32_t num_actor_layers, int32_t num_critic_layers) {
  if (!owner || !other || owner == other) return fail("ShareParameters needs two distinct learners");
  if (owner->cfg.device != other->cfg.device) return fail("ShareParameters: both learners must live on the same device");
  if (!same_nets(owner, other)) return fail("ShareParameters: net shapes differ");
  if (owner->fp16 || other->fp16) return fail("ShareParameters is not supported in fp16 mode (each learner keeps private fp16 weight copies)");
  if (owner->w_owner) return fail("ShareParameters: the owner itself shares another learner's layers; share from the root");
  if (other->w_owner && other->w_owner != owner) return fail("ShareParameters: already sharing with a different owner");
  size_t fa = 0, fc = 0;
  RC(shared_prefix(owner->la, num_actor_layers, &fa));
  RC(shared_prefix(owner->lc, num_critic_layers, &fc));
  HIPCHK(hipSetDevice(owner->cfg.device));
  HIPCHK(hipStreamSynchronize(owner->stream));
  HIPCHK(hipStreamSynchronize(other->stream));
  if (!other->w_owner && (fa || fc)) owner->sharers += 1;
  if (other->w_owner && !(fa || fc)) owner->sharers -= 1;
  other->w_owner = (fa || fc) ? owner : nullptr;
  other->shared_fl[0] = fa; other->shared_fl[1] = fc;
  drop_graphs(other);                      // captured launches hold the old weight pointers
  return 0;
}

int dqnhip_share_replay_memory(dqnhip_handle owner, dqnhip_handle other) {
  if (!owner || !other || owner == other) return fail("ShareReplayMemory needs two distinct learners");
  if (owner->cfg.device != other->cfg.device) return fail("ShareReplayMemory: both learners must live on the same device");
  if (owner->S != other->S) return fail("ShareReplayMemory: state sizes differ");
  H* root = RO(owner);
  if (RO(other) == root) return 0;
  if (other->sharers && other->ring_shared) return fail("ShareReplayMemory: other learners already use this learner's memory");
  HIPCHK(hipSetDevice(owner->cfg.device));
  HIPCHK(hipStreamSynchronize(owner->stream));
  HIPCHK(hipStreamSynchronize(other->stream));
  if (other->ring_owner) other->ring_owner->sharers -= 1;
  if (!root->ring_ev) HIPCHK(hipEventCreateWithFlags(&root->ring_ev, hipEventDisableTiming));
  root->ring_shared = true;
  root->sharers += 1;
  other->ring_owner = root;                // other's deque is dropped: shared_ptr assignment, src/dqn.cpp:1081
  drop_graphs(other);
  return 0;
}

int dqnhip_get_iters(dqnhip_handle h, int32_t* actor_iter, int32_t* critic_iter) {
  if (!h) return fail("null handle");
  if (actor_iter) *actor_iter = h->h_actor_iter;
  if (critic_iter) *critic_iter = h->h_critic_iter;
  return 0;
}

int dqnhip_set_iters(dqnhip_handle h, int32_t actor_iter, int32_t critic_iter) {
  if (!h) return fail("null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  int v[2] = {actor_iter, critic_iter};
  HIPCHK(hipMemcpy(&h->st->actor_iter, v, sizeof v, hipMemcpyHostToDevice));
  h->h_actor_iter = actor_iter; h->h_critic_iter = critic_iter;
  return 0;
}

// ---- introspection ----------------------------------------------------------------------

int dqnhip_debug_read(dqnhip_handle h, const char* name, float* host, size_t count) {
  if (!h || !name || !host) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t B = h->B;
  const float* src = nullptr; size_t n = B; bool pad16 = false; bool is_int = false;
  if (!strcmp(name, "q_target")) src = h->q_t;
  else if (!strcmp(name, "y")) src = h->y;
  else if (!strcmp(name, "q_train")) src = h->q1;
  else if (!strcmp(name, "q_policy")) src = h->q2;
  else if (!strcmp(name, "terminal")) src = h->mb_term;
  else if (!strcmp(name, "actor_out")) { src = h->aout16; pad16 = true; n = B * kNO; }
  else if (!strcmp(name, "dq_da")) { src = h->dA16; pad16 = true; n = B * kNO; }
  else if (!strcmp(name, "idx")) { src = (const float*)h->mb_idx; is_int = true; }
  else if (!strncmp(name, "act", 3) && name[3] >= '0' && name[3] <= '4' && name[4] == '_') {
    // "act<p>_<i>": the stored (post-ReLU, in place: src/dqn.cpp:409-410) tower activations of the last update's pass p
    // (0 actor_target(s'), 1 actor(s), 2 critic_target, 3 critic(s, a), 4 critic(s, mu(s))), layer i = 1 .. L, dense
    // [B][width].  Parity tests compare their SIGNS with the oracle's: an fp32 evaluation may put a pre-activation that
    // is within round-off of zero on the other side, which switches that unit's ReLU' between 1 and 0.01 for that row.
    const int p = name[3] - '0', i = atoi(name + 5);
    const NetLayout& l = layout_of(h, p >= 2);
    if (i < 1 || i > h->L) return fail("debug buffer '%s': layer out of range", name);
    const size_t W = l.dims[i];
    if (count < B * W) return fail("buffer too small for '%s': %zu < %zu", name, count, B * W);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->fp16) {
      std::vector<h16> t16(B * W);
      HIPCHK(hipMemcpy(t16.data(), h->act16[p][i], t16.size() * sizeof(h16), hipMemcpyDeviceToHost));
      for (size_t e = 0; e < t16.size(); ++e) host[e] = (float)t16[e];
    } else HIPCHK(hipMemcpy(host, h->act[p][i], B * W * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  else return fail("unknown debug buffer '%s'", name);
  if (count < n) return fail("buffer too small for '%s': %zu < %zu", name, count, n);
  std::vector<float> tmp(pad16 ? B * kAP : B);
  HIPCHK(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (pad16) { for (size_t r = 0; r < B; ++r) for (int c = 0; c < kNO; ++c) host[r * kNO + c] = tmp[r * kAP + c]; }
  else if (is_int) { for (size_t r = 0; r < B; ++r) host[r] = (float)reinterpret_cast<const int*>(tmp.data())[r]; }
  else memcpy(host, tmp.data(), B * sizeof(float));
  return 0;
}

int dqnhip_get_stream(dqnhip_handle h, void** stream) {
  if (!h || !stream) return fail("null argument");
  *stream = (void*)h->stream;
  return 0;
}

int dqnhip_set_kernel_timing(dqnhip_handle h, int32_t enable) {
  if (!h) return fail("null handle");
  h->timing = enable != 0;
  return 0;
}

int dqnhip_get_kernel_timing(dqnhip_handle h, const char* family, float* avg_ms, int64_t* launches, int32_t reset) {
  if (!h || !family) return fail("null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  int fam = -1;
  for (int i = 0; i < kNumFamily; ++i) if (!strcmp(family, kFamily[i])) fam = i;
  if (fam < 0) return fail("unknown kernel family '%s' (gemm_fwd_lds_4x2|gemm_fwd_lds_2x2|gemm_fwd_direct|gemm_dgrad|gemm_wgrad|gemm_bwd_pair|adam|hgemm_fwd|hgemm_dgrad|hgemm_wgrad)", family);
  HIPCHK(hipStreamSynchronize(h->stream));
  double total = 0; int64_t cnt = 0;
  for (auto& r : h->recs) {
    if (r.family != fam) continue;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    total += ms; cnt += 1;
  }
  if (avg_ms) *avg_ms = cnt ? (float)(total / cnt) : 0.0f;
  if (launches) *launches = cnt;
  if (reset) {
    for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    h->recs.clear();
  }
  return 0;
}


// ---- batched env front-end (include/dqnhip_env.h) -----------------------------------------
}  // extern "C"

struct dqnhip_env {
  dqnhip_learner* h = nullptr;
  dqnhip_env_config cfg{};
  EnvDev d{};
  int Npad = 0;
  float* acts[kMaxL + 1] = {nullptr};
  std::vector<void*> allocs;
  float* eps_dev = nullptr;
  int* commit_ticket = nullptr;
  hipGraphExec_t graph[2] = {nullptr, nullptr};   // one batched step / kEnvUnroll steps, captured on first use
  bool graph_failed = false;
  // inside a sequence of batched steps the episode flush of step t (LabelTransitions + AddTransitions of the
  // finished episodes) rides as extra workgroups of step t+1's first-layer launch: k_env_step resets the worker
  // itself, so nothing before the next k_env_step depends on the flush
  bool flush_deferred = false;
};
constexpr int kEnvUnroll = 16;

namespace {
template <typename T>
int env_alloc(dqnhip_env* e, T** p, size_t n) {
  HIPCHK(hipMalloc(p, n * sizeof(T)));
  HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(T), e->h->stream));
  e->allocs.push_back((void*)*p);
  return 0;
}
}  // namespace

extern "C" {

static int env_create_impl(dqnhip_env* e);

int dqnhip_env_create(dqnhip_handle h, const dqnhip_env_config* cfg, dqnhip_env_handle* out) {
  if (!h || !cfg || !out) return fail("null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(dqnhip_env_config)) return fail("dqnhip_env_config.struct_size mismatch");
  if (cfg->workers < 1 || cfg->workers > (1 << 20)) return fail("workers out of range");
  if (cfg->max_steps < 1 || cfg->max_steps > 4096) return fail("max_steps out of range");
  if (h->S < 56) return fail("HFOGameState reads state indices up to 55: state_size must be >= 56 (src/hfo_game.cpp:130-152)");
  if ((long long)cfg->workers * cfg->max_steps >= RO(h)->ring.cap)
    return fail("replay capacity %d must exceed workers*max_steps = %lld", RO(h)->ring.cap, (long long)cfg->workers * cfg->max_steps);
  HIPCHK(hipSetDevice(h->cfg.device));
  dqnhip_env* e = new dqnhip_env();
  e->h = h; e->cfg = *cfg;
  const int rc = env_create_impl(e);
  if (rc) { const std::string msg = g_err; dqnhip_env_destroy(e); g_err = msg; return rc; }
  *out = e;
  return 0;
}

static int env_create_impl(dqnhip_env* e) {
  dqnhip_learner* h = e->h;
  const dqnhip_env_config* cfg = &e->cfg;
  EnvDev& d = e->d;
  d.N = cfg->workers; d.S = h->S; d.SP = h->la.kp[0]; d.T = cfg->max_steps; d.unum = cfg->unum;
  d.p_end = cfg->p_end; d.p_goal = cfg->p_goal; d.seed = cfg->seed;
  e->Npad = round_up(d.N, 32);
  const size_t N = d.N, Np = e->Npad;
  RC(env_alloc(e, &d.cur, Np * d.SP)); RC(env_alloc(e, &d.out16, Np * kAP));
  RC(env_alloc(e, &d.ep_s, N * d.T * d.SP)); RC(env_alloc(e, &d.ep_a, N * d.T * kAP)); RC(env_alloc(e, &d.ep_r, N * d.T));
  RC(env_alloc(e, &d.game, N)); RC(env_alloc(e, &d.len, N)); RC(env_alloc(e, &d.done, N)); RC(env_alloc(e, &d.g, N));
  RC(env_alloc(e, &d.act, N)); RC(env_alloc(e, &d.arg1, N)); RC(env_alloc(e, &d.arg2, N)); RC(env_alloc(e, &d.rew, N));
  RC(env_alloc(e, &d.n_steps, N)); RC(env_alloc(e, &d.n_episodes, N)); RC(env_alloc(e, &d.n_goals, N)); RC(env_alloc(e, &d.reward_sum, N));
  e->acts[0] = d.cur;
  for (int i = 1; i <= h->L; ++i) RC(env_alloc(e, &e->acts[i], Np * h->la.kp[i]));
  RC(env_alloc(e, &e->eps_dev, 16)); d.eps = e->eps_dev;
  RC(env_alloc(e, &e->commit_ticket, 32));
  hipLaunchKernelGGL(k_env_init, dim3(d.N), dim3(64), d.SP * sizeof(float), h->stream, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int dqnhip_env_destroy(dqnhip_env_handle e) {
  if (!e) return 0;
  hipSetDevice(e->h->cfg.device);
  hipStreamSynchronize(e->h->stream);
  for (void* p : e->allocs) hipFree(p);
  for (int i = 0; i < 2; ++i) if (e->graph[i]) hipGraphExecDestroy(e->graph[i]);
  delete e;
  return 0;
}

// First tower layer of batched step t+1 (the small-K direct kernel's 32x32 tiles) and the episode flush of step t
// in ONE launch: blocks [0, tiles) are GEMM tiles, the next N blocks are k_env_flush's.  The two parts share no data
// (the layer reads the state panel k_env_step(t) wrote, the flush reads done[] and the episode rows).  A side
// stream was measured first (A/B in one call, 64 workers, S = 68): 48.0 us per step against 33.5 — every
// cross-stream edge of a replayed graph costs more than the 7 us flush it would hide.
__global__ __launch_bounds__(256) void k_env_l0_flush(const GemmBatch batch, EnvDev e, Ring ring, const DevState* st, double gamma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < batch.total_tiles) {
    int pi, tile_p, tile_q;
    tile_of_block(batch, pi, tile_p, tile_q);
    fwd_direct_body<2, 2>(batch.prob[pi], tile_p, tile_q, smem);
    return;
  }
  env_flush_block(e, ring, st, gamma, (int)blockIdx.x - batch.total_tiles, (int)gridDim.x - batch.total_tiles, smem);
}
// one batched env step on the learner's stream: SelectActionGreedily for all workers, then the
// per-worker epsilon draw / GetAction / reward / episode bookkeeping / AddTransitions
static int env_one_step(dqnhip_env* e, bool more_follow) {
  dqnhip_learner* h = e->h;
  EnvDev d = e->d;
  hipStream_t st = h->stream;
  const NetLayout& la = h->la;
  FwdPass fp{DQNHIP_ACTOR, &la, e->acts};
  // 5 launches per batched step at L = 4 inside a sequence: the actor heads ride in k_env_step (the four waves of a
  // worker's block compute its own 10 outputs), the ring bookkeeping in the flush's last block, and the flush itself
  // in the NEXT step's first-layer launch.
  // (beyond a few hundred workers the dedicated head kernel and a separate commit win: one block per head row is
  // slower than the tiled head kernel there, and N arrivals on one counter serialise at ~12 ns each)
  const bool fused = la.dims[la.L] % 4 == 0 && d.N <= 512;
  if (fused) {
    d.head_x = e->acts[la.L]; d.head_h = la.dims[la.L];
    d.head_w = wat(h, DQNHIP_ACTOR, la.hw_off); d.head_b = wat(h, DQNHIP_ACTOR, la.hb_off);
    d.commit_ticket = e->commit_ticket;
  }
  const bool l0_direct = !((la.kp[0] >= 512) && (la.kp[0] % 256 == 0)) && la.dims[1] % 32 == 0 && e->Npad % 32 == 0;
  int first = 0;
  if (e->flush_deferred) {
    // the previous step's flush + this step's first layer (the deferral below is only made when this holds)
    GemmBatch b{}; b.n = 1;
    GemmProblem& p = b.prob[0];
    p.P = wat(h, DQNHIP_ACTOR, la.w_off[0]); p.ldp = la.kp[0];
    p.Q = e->acts[0]; p.ldq = la.kp[0];
    p.C = e->acts[1]; p.ldc = la.kp[1];
    p.Pdim = la.dims[1]; p.Qdim = e->Npad; p.Kred = la.kp[0];
    p.bias = wat(h, DQNHIP_ACTOR, la.b_off[0]); p.relu = 1;
    p.tiles_p = p.Pdim / 32; p.tiles_q = p.Qdim / 32; p.tile_base = 0;
    b.total_tiles = p.tiles_p * p.tiles_q;
    const size_t lds = std::max<size_t>(4 * 2 * 2 * 64 * 16, d.T * sizeof(float));
    hipLaunchKernelGGL(k_env_l0_flush, dim3(b.total_tiles + d.N), dim3(256), lds, st, b, d, RO(h)->ring,
                       (const DevState*)RO(h)->st, h->cfg.gamma);
    HIPCHK(hipGetLastError());
    e->flush_deferred = false;
    first = 1;
  }
  for (int i = first; i < la.L; ++i) RC(layer_forward(h, st, &fp, 1, e->Npad, i));
  if (!fused) {
    HeadArgs a{}; a.X = e->acts[la.L]; a.ldx = la.dims[la.L]; a.H = la.dims[la.L]; a.rows = e->Npad;
    a.W = wat(h, DQNHIP_ACTOR, la.hw_off); a.b = wat(h, DQNHIP_ACTOR, la.hb_off); a.out16 = d.out16;
    RC((head_forward<kNO, HEAD_ACTOR>(h, st, a)));
  }
  hipLaunchKernelGGL(k_env_step, dim3(d.N), dim3(256), 2 * d.SP * sizeof(float), st, d);
  HIPCHK(hipGetLastError());
  if (more_follow && fused && l0_direct && !h->timing) { e->flush_deferred = true; return 0; }
  hipLaunchKernelGGL(k_env_flush, dim3(d.N), dim3(256), d.T * sizeof(float), st, d, RO(h)->ring,
                     (const DevState*)RO(h)->st, h->cfg.gamma);
  HIPCHK(hipGetLastError());
  if (d.commit_ticket == nullptr) {
    hipLaunchKernelGGL(k_env_commit, dim3(1), dim3(256), 0, st, d, RO(h)->ring, RO(h)->st);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

static int env_capture(dqnhip_env* e, int which) {
  dqnhip_learner* h = e->h;
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  const int n = which ? kEnvUnroll : 1;
  for (int s = 0; s < n && !rc; ++s) rc = env_one_step(e, s + 1 < n);
  e->flush_deferred = false;                     // (only left set if a launch failed: the sequence is abandoned)
  hipError_t err = hipStreamEndCapture(h->stream, &graph);
  if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
  if (err != hipSuccess) return fail("hipStreamEndCapture (env): %s", hipGetErrorString(err));
  err = hipGraphInstantiate(&e->graph[which], graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (err != hipSuccess) return fail("hipGraphInstantiate (env): %s", hipGetErrorString(err));
  return 0;
}

int dqnhip_env_step(dqnhip_env_handle e, float epsilon, int32_t n_steps) {
  if (!e) return fail("null env");
  if (!(epsilon >= 0.0f && epsilon <= 1.0f)) return fail("Check failed: epsilon >= 0.0 && epsilon <= 1.0");   // src/dqn.cpp:698
  if (n_steps < 1) return fail("n_steps must be >= 1");
  dqnhip_learner* h = e->h;
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  RingUse ring_use(h);
  hipLaunchKernelGGL(k_set_float<0>, dim3(1), dim3(1), 0, st, e->eps_dev, epsilon);
  HIPCHK(hipGetLastError());
  // the step is a fixed launch sequence (9 launches at L = 4, ~6 us each when launch-bound): replay it
  // as a hipGraph unless the learner's layers may be re-pointed (sharing) or graphs are off
  const bool use_graph = h->cfg.use_graph && !e->graph_failed && !h->timing && !h->w_owner && !h->ring_owner;
  int s = 0;
  if (use_graph) {
    for (int which = 1; which >= 0; --which) {
      const int n = which ? kEnvUnroll : 1;
      while (n_steps - s >= n) {
        if (!e->graph[which] && env_capture(e, which)) { e->graph_failed = true; break; }
        HIPCHK(hipGraphLaunch(e->graph[which], st));
        s += n;
      }
      if (e->graph_failed) break;
    }
  }
  for (; s < n_steps; ++s) {
    const int rc = env_one_step(e, s + 1 < n_steps);
    if (rc) { e->flush_deferred = false; return rc; }
  }
  RO(h)->ring_stale = true;
  return 0;
}

int dqnhip_env_stats(dqnhip_env_handle e, int64_t* env_steps, int64_t* episodes, double* reward_sum, int64_t* goals) {
  if (!e) return fail("null env");
  dqnhip_learner* h = e->h;
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t N = e->d.N;
  std::vector<unsigned long long> a(N), b(N), c(N); std::vector<double> r(N);
  HIPCHK(hipMemcpyAsync(a.data(), e->d.n_steps, N * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(b.data(), e->d.n_episodes, N * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(c.data(), e->d.n_goals, N * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(r.data(), e->d.reward_sum, N * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  long long s0 = 0, s1 = 0, s2 = 0; double s3 = 0;
  for (size_t i = 0; i < N; ++i) { s0 += a[i]; s1 += b[i]; s2 += c[i]; s3 += r[i]; }
  if (env_steps) *env_steps = s0; if (episodes) *episodes = s1; if (goals) *goals = s2; if (reward_sum) *reward_sum = s3;
  RingUse ring_use(h);
  return refresh_ring(h);
}

int dqnhip_env_debug_read(dqnhip_env_handle e, const char* name, float* host, size_t count) {
  if (!e || !name || !host) return fail("null argument");
  dqnhip_learner* h = e->h;
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t N = e->d.N;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (!strcmp(name, "action") || !strcmp(name, "episode_len")) {
    if (count < N) return fail("buffer too small");
    std::vector<int> t(N);
    HIPCHK(hipMemcpy(t.data(), !strcmp(name, "action") ? e->d.act : e->d.len, N * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) host[i] = (float)t[i];
    return 0;
  }
  const float* src = nullptr; size_t n = N;
  if (!strcmp(name, "arg1")) src = e->d.arg1;
  else if (!strcmp(name, "arg2")) src = e->d.arg2;
  else if (!strcmp(name, "reward")) src = e->d.rew;
  else if (!strcmp(name, "state")) {
    if (count < N * h->S) return fail("buffer too small");
    std::vector<float> t(N * e->d.SP);
    HIPCHK(hipMemcpy(t.data(), e->d.cur, t.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) memcpy(host + i * h->S, &t[i * e->d.SP], h->S * 4);
    return 0;
  } else if (!strcmp(name, "actor_out")) {
    // the ActorOutput chosen at the last step = last written row of the open episode, or (if the
    // episode just ended) not available any more: report the greedy output instead
    if (count < N * kNO) return fail("buffer too small");
    std::vector<float> t(N * kAP);
    HIPCHK(hipMemcpy(t.data(), e->d.out16, t.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i) memcpy(host + i * kNO, &t[i * kAP], kNO * 4);
    return 0;
  } else return fail("unknown env debug buffer '%s'", name);
  if (count < n) return fail("buffer too small");
  HIPCHK(hipMemcpy(host, src, n * 4, hipMemcpyDeviceToHost));
  return 0;
}
}  // extern "C"
