// learner_internal.hip.h — what the translation units of libdqnhip.so share: the learner object behind dqnhip_handle, the
// parameter-arena layout, the error convention, and the prototypes of the host-side building blocks.
//   learner.hip      the update (forward / backward building blocks, the three phases, graph capture, the update entry points)
//   learner_dp.hip   native data parallelism: RCCL inside the library, the file rendezvous (dqnhip_dp_*)
//   learner_io.hip   acting, replay memory (+ .replaymemory files), parameters, multi-agent sharing, introspection
//   learner_env.hip  host side of the batched env front-end (include/dqnhip_env.h)
//   snapshot.cpp     Caffe snapshot layout (no device code)
// Not installed, not part of the C-ABI.
#pragma once
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


namespace dqnhip_host {
using namespace dqnhip;

extern thread_local std::string g_err;      // defined in learner.hip; dqnhip_last_error() returns it

inline int fail(const char* fmt, ...) {
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

void layout_init(NetLayout& l, int in_dim, const dqnhip_config& c, bool actor);

struct TimingRec { int family; hipEvent_t a, b; };

}  // namespace dqnhip_host

using namespace dqnhip;          // (internal header: every includer is a translation unit of this library)
using namespace dqnhip_host;

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
  int rccl_version = 0;                 // ncclGetVersion of the build this process resolved (cross-checked over the group in dqnhip_dp_init)
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
  // inside multi-update graphs that gather one launch group early (early_l0): the two panels an update still reads after the next
  // update's gather has run exist twice, by update parity; Xa_s / Xc_pl (and act[1][0] / act[4][0]) point at the current update's
  float* Xa_s2[2] = {nullptr, nullptr}; float* Xc_pl2[2] = {nullptr, nullptr};
  float* act[5][kMaxL + 1] = {{nullptr}};
  float* dZa[kMaxL + 1] = {nullptr};
  float* dZc[kMaxL + 1] = {nullptr};
  float *mb_reward = nullptr, *mb_mc = nullptr, *mb_term = nullptr;
  float* qdot[2] = {nullptr, nullptr};   // [B][kp[L] / 16]: the head dot products of critic_target(s', .) / critic(s, a) in 16-column pieces (GemmProblem::dot_w)
  float* Wact_t = nullptr;              // [kNO][dims[1]] (first_layers_launch)
  float* Zs = nullptr;                  // [B][kp[1]]: the state half of critic_target's first layer, before bias / ReLU (first_layers_launch)
  float* U3 = nullptr;                  // [B][kp[L]]: (-w_h) lrelu'(x_L) of the critic(s, a) training pass (left by its top forward layer for k_dgrad_qtrain)
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
  int cap_n = 16;              // ... and the number of updates of that graph (the last one carries no riders for a successor)
  // dqnhip_update_async_n: graphs of 8 / 4 / 2 updates for what is left after the sixteen-update graphs (round 6: a remainder of r
  // updates used to be r one-update graphs, each with its own gather, its own first-layer launch and the 8-us gap between two graph
  // launches — the driver's 20 timed steps are 16 + 4)
  hipGraphExec_t graph_small[3] = {nullptr, nullptr, nullptr};
  // [0]: device-sampled, [1]: explicit idx (pinned buffer), [2], [3]: explicit idx in the pipelined slots, [4]: kMultiU device-sampled
  // updates (dqnhip_update_async_n), [5] .. [7]: dqnhip_update_chained — head of a chain (own gather, parity 0), continued at parity 1 / 0
  hipGraphExec_t graph_exec[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool graph_failed = false;
  // dqnhip_update_chained (the drop-in's blocking update with the NEXT update's indices known one call ahead): the next update's gather
  // and first layers ride in this update's optimiser launches, as inside a multi-update graph.  What rode along is only used if the
  // next call brings exactly those indices and nothing touched weights, iterations or the replay memory in between (epoch).
  unsigned long long epoch = 0;         // bumped by every entry point that changes weights, iteration counters or the replay memory
  bool chain_cap = false;               // a chain graph is being captured (gather_args: riders read idx_next, counters = live + 1)
  bool chain_valid = false; int chain_par = 0;
  unsigned long long chain_epoch = 0, chain_ring_epoch = 0;
  std::vector<int32_t> chain_idx;       // the indices the riders gathered
  int* idx_next_pinned[2] = {nullptr, nullptr}; const int* idx_next_dev[2] = {nullptr, nullptr};   // by the parity of the update they belong to
  // dqnhip_update_pipelined
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};
  int* pipe_idx_dev[2] = {nullptr, nullptr}; int* pipe_idx_pinned[2] = {nullptr, nullptr};
  float* pipe_stats[2] = {nullptr, nullptr};
  unsigned long long pipe_count = 0;
};

namespace dqnhip_host {

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

extern const char* const kFamily[];
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
void dense_to_arena(const NetLayout& l, const float* dense, std::vector<float>& arena);
void arena_to_dense(const NetLayout& l, const std::vector<float>& arena, float* dense);
inline const NetLayout& layout_of(const H* h, int net) { return (net & 1) ? h->lc : h->la; }
int validate(const dqnhip_config* c);
inline size_t grad_arena_floats(const NetLayout& la, const NetLayout& lc) { return la.arena + 64 + lc.arena + 64; }

// ---- building blocks defined in learner.hip, used by the other translation units ---------------------------------
// seed_w / seed_out: the TOP layer's launch also writes the dq = -1 pass's tower-top gradient (GemmProblem::seed_w)
struct FwdPass { int net; const NetLayout* l; float** act; const float* seed_w = nullptr; float* seed_out = nullptr;
                 const float* dot_w = nullptr; float* dot_out = nullptr; };   // dot_w / dot_out: GemmProblem::dot_w, same layer

int layer_forward(H* h, hipStream_t st, const FwdPass* passes, int n, int rows, int i);
int tower_forward(H* h, hipStream_t st, const FwdPass* passes, int n, int rows, int first_layer = 0);   // first_layer 1: layer 0 came out of a FirstLayerRider
template <int NH, int MODE>
int head_forward(H* h, hipStream_t st, const HeadArgs& a, const HeadArgs* b = nullptr) {
  HeadArgs2 a2{}; a2.p[0] = a; if (b) a2.p[1] = *b;
  if (NH > 1 && a.rows >= 1024 && a.H <= 1024 && a.H % 4 == 0)   // (single-head: the block-per-row form measured faster, 6.6 vs 8.8 us)
    hipLaunchKernelGGL((k_head_fwd_rows<NH, MODE>), dim3(256, b ? 2 : 1), dim3(256), 0, st, a2);
  else if (a.l1_y != nullptr || (b && b->l1_y != nullptr))      // Step(1): the target actor's head also finishes critic_target's first layer
    hipLaunchKernelGGL((k_head_fwd<NH, MODE, true>), dim3(std::min(a.rows, 1024), b ? 2 : 1), dim3(256), 0, st, a2);
  else
    hipLaunchKernelGGL((k_head_fwd<NH, MODE, false>), dim3(std::min(a.rows, 1024), b ? 2 : 1), dim3(256), 0, st, a2);
  HIPCHK(hipGetLastError());
  return 0;
}
constexpr int kMultiU = 16;    // updates per replay of the multi-update graph (dqnhip_update_async_n; see capture_graph)
// Philox key of SampleTransitionsFromMemory: cfg.seed on rank 0 (what oracle/c_oracle.philox_indices
// reproduces); data-parallel ranks get distinct streams from the SAME cfg.seed, so that the weight
// initialisation (also keyed by cfg.seed) stays identical across the group
inline uint64_t sample_key(const H* h) { return (uint64_t)h->cfg.seed + 0x9E3779B97F4A7C15ull * (uint64_t)h->cfg.dp_rank; }
inline void shard_range(const H* h, int net, size_t& lo, size_t& hi, int rank = -1) {
  const size_t slice = layout_of(h, net).arena / (size_t)h->cfg.dp_world;
  const size_t r = (size_t)(rank < 0 ? h->cfg.dp_rank : rank);
  lo = r * slice; hi = lo + slice;
}
// Which merged forms the update of a learner takes (plan_of, learner.hip: the one place that decides; dqnhip_get_update_plan reports it)
struct UpdatePlan {
  bool fp16, dp;
  bool shifted_c, shifted_a;        // the shifted backward schedule (tower_backward) for the critic's Step(1) / the actor's backward
  bool head_rides_c, head_rides_a;  // the head's dW / db as rider blocks of the net's last backward launch
  bool fuse_q;                      // k_dgrad_qtrain: Step(1)'s head arithmetic inside the critic's top-layer dgrad launch
  bool fused_seed;                  // the dq = -1 seed from the top layer's forward epilogue, q(s, mu(s)) as rider blocks
  bool fuse_head;                   // k_dqda_head_bwd: dQ/da's last step + inverting gradients + actor heads' backward in one launch
  bool critic_l0;                   // the first layer of critic(s, mu(s)) inside the critic's optimiser launch
  bool first_layers_merged;         // Step(1)'s four first layers in one launch (critic_target's action half in the head kernel)
  bool early_l0;                    // multi-update graphs: next gather in the critic's optimiser launch, next first layers in the actor's
  bool tails_ride;                  // data parallel: the [loss, q, flag] tails block rides in each net's last backward launch (no k_tails launches)
};
UpdatePlan plan_of(const H* h);
// the current update's copies of the two double-buffered panels (by update parity inside a multi-update graph that gathers early;
// parity 0 everywhere else — every capture ends by selecting parity 0 again)
inline void select_panels(H* h, int par) { h->Xa_s = h->Xa_s2[par]; h->Xc_pl = h->Xc_pl2[par]; h->act[1][0] = h->Xa_s; h->act[4][0] = h->Xc_pl; }
struct NextL0 { ActorL0 a; PlainL0 c, ct; };           // the next update's first layers as riders of the actor's optimiser launch
int adam_launch(H* h, hipStream_t st, int net, const float* partial, int n_partial, size_t begin, size_t end, const TickArgs* tick = nullptr,
                bool corr_pre = true, const FirstLayerRider* fl = nullptr, const GatherArgs* early_gather = nullptr, const NextL0* next_l0 = nullptr);
int sumsq_launch(H* h, int net, size_t begin = 0, size_t end = 0);
int to_bf16_launch(H* h, int net);                       // k_to_bf16: the bf16 transfer image of a gradient arena
int shard_scal_launch(H* h, float* tail);                // k_shard_scal: this rank's share of the clip norm -> tail[3]
int run_phase(H* h, int phase, const int* idx_dev);
int sync_dirty16(H* h);
int refresh_ring(H* h);
int stage_indices(H* h, const int32_t* idx_host, const int** idx_dev);
int ensure_stage(H* h, size_t bytes);
int ensure_act(H* h, int rows);
void drop_graphs(H* h);                                  // every captured launch sequence of this learner
// learner_dp.hip
int dp_reduce_slice(H* h, hipStream_t st, int net, size_t off, size_t count);
int dp_allgather_weights(H* h, int net);
int dp_destroy_impl(H* h, bool keep_learner);
const char* rccl_path();
// learner_io.hip
bool same_nets(const H* a, const H* b);

}  // namespace dqnhip_host
