// learner_env.hip — host side of the batched env front-end (include/dqnhip_env.h; device side: env.hip.h).
#include "learner_internal.hip.h"

using namespace dqnhip;
using namespace dqnhip_host;

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
static __global__ __launch_bounds__(256) void k_env_l0_flush(const GemmBatch batch, EnvDev e, Ring ring, const DevState* st, double gamma) {
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
  RO(h)->epoch += 1;            // episodes may end inside: the ring changes on the device
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

