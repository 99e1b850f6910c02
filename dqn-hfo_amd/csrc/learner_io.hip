// learner_io.hip — everything of the C-ABI around the update: acting (SelectActions / CriticForward), the replay memory and
// its .replaymemory files, parameter access, multi-agent sharing, introspection (include/dqnhip.h).
#include "learner_internal.hip.h"

using namespace dqnhip;
using namespace dqnhip_host;

extern "C" {

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
  RO(h)->epoch += 1;          // (dqnhip_update_chained: whatever a rider gathered ahead is stale)
  // n == 0 is AddTransitions of an empty vector: `while (size() + 0 >= capacity) pop_front()` (src/dqn.cpp:776) still evicts
  // one transition from a full deque — only the bookkeeping runs
  if (n < 0 || (n == 0 && single != 0)) return fail("n must be >= 1");
  RingUse ring_use(h);
  RC(refresh_ring(h));
  const long long cap = RO(h)->ring.cap;
  if (single == 0 && n >= cap) return fail("AddTransitions: batch of %d does not fit capacity %lld (the reference would pop an empty deque)", n, cap);
  if (single == 2 && RO(h)->h_size + n > cap) return fail("LoadReplayMemory: %lld transitions exceed the capacity %lld", RO(h)->h_size + n, cap);
  hipLaunchKernelGGL(k_add_transitions, dim3(std::max(1, (n + 3) / 4)), dim3(256), 0, h->stream, RO(h)->ring, RO(h)->st, s, a, r, mc, nx,
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
  if (n < 0 || (n == 0 && single != 0)) return fail("n must be >= 1");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (n == 0) return add_dev(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0);      // empty AddTransitions: eviction only
  if (!s || !a || !r || !mc || !term) return fail("null input array");
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
  RO(h)->epoch += 1;
  HIPCHK(hipSetDevice(h->cfg.device));
  RingUse ring_use(h);
  HIPCHK(hipMemsetAsync(RO(h)->st, 0, 2 * sizeof(int), h->stream));   // ring_head, ring_size
  RO(h)->h_head = 0; RO(h)->h_size = 0; RO(h)->ring_stale = false;
  return 0;
}


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
  RO(h)->epoch += 1;
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
  // sharded optimiser: this rank holds the Adam history of its own slice only until the group has gathered it — handing out
  // (or snapshotting: snapshot.cpp reads m, v through here) the stale rest would poison a later resume silently
  if (h->dp_shard && h->shard_stale && (kind == DQNHIP_KIND_M || kind == DQNHIP_KIND_V))
    return fail("dqnhip_get_params: the optimiser is sharded and updates ran since the last dqnhip_dp_gather_state — this rank holds the Adam "
                "history of its own slice only; call dqnhip_dp_gather_state on every rank first");
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
  h->epoch += 1;
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
  h->epoch += 1;
  if (net != DQNHIP_ACTOR && net != DQNHIP_CRITIC) return fail("net must be ACTOR or CRITIC");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t sh = h->shared_fl[net], n = layout_of(h, net).arena;
  if (sh) HIPCHK(hipMemcpyAsync(h->w_owner->w[net + 2], h->w_owner->w[net], sh * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  if (sh < n) HIPCHK(hipMemcpyAsync(h->w[net + 2] + sh, h->w[net] + sh, (n - sh) * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  h->w16_dirty[net + 2] = true;
  return 0;
}

// ---- multi-agent sharing (src/dqn.cpp:1036-1083, src/dqn_main.cpp:305-323) ------------------

}  // extern "C"
namespace dqnhip_host {
bool same_nets(const H* a, const H* b) {
  if (a->S != b->S || a->L != b->L) return false;
  for (int i = 0; i < a->L; ++i) if (a->cfg.hidden[i] != b->cfg.hidden[i]) return false;
  return true;
}
}  // namespace dqnhip_host
extern "C" {

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
  owner->epoch += 1; other->epoch += 1;
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
  owner->epoch += 1; other->epoch += 1; RO(owner)->epoch += 1;
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
  h->epoch += 1;
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

}  // extern "C"
