// learner_dp.hip — native data parallelism of the learner: RCCL over xGMI inside libdqnhip.so (include/dqnhip.h dqnhip_dp_*;
// SURVEY 8e).  No device code of its own: the kernels it needs are launched through learner.hip's helpers.
#include "learner_internal.hip.h"

using namespace dqnhip;
using namespace dqnhip_host;
#include <dlfcn.h>

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


namespace dqnhip_host {
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
      RC(to_bf16_launch(h, net));
      NCCLCHK(ncclReduceScatter(h->g16[net], h->g16[net] + lo, hi - lo, ncclBfloat16, ncclSum, h->comm, st));
    } else {
      NCCLCHK(ncclReduceScatter(h->g[net], h->g[net] + lo, hi - lo, ncclFloat, ncclSum, h->comm, st));
    }
    RC(sumsq_launch(h, net, lo, hi));
    RC(shard_scal_launch(h, tail));
    NCCLCHK(ncclAllReduce(tail, tail, 4, ncclFloat, ncclSum, h->comm, st));
    return 0;
  }
  if (h->dp_half) {
    // bf16 image of the arena -> sum all-reduce -> (phase 1 / 2 widen it again inside k_sumsq_bf16).  The fp32
    // tails of both nets travel once, with the actor's gradients (nothing reads them before the tick of phase 2).
    RC(to_bf16_launch(h, net));
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
  select_panels(h, 0);
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

// the file the running process resolved RCCL from
const char* rccl_path() {
  Dl_info info{};
  if (dladdr(reinterpret_cast<const void*>(&ncclGetVersion), &info) && info.dli_fname) return info.dli_fname;
  return "?";
}
}  // namespace dqnhip_host

extern "C" {

/* RCCL version (ncclGetVersion) and the shared object it was resolved from in THIS process; needs no communicator and no GPU. */
int dqnhip_dp_info(int32_t* rccl_version, char* path, size_t path_bytes) {
  int ver = 0;
  NCCLCHK(ncclGetVersion(&ver));
  if (rccl_version) *rccl_version = ver;
  if (path && path_bytes) { snprintf(path, path_bytes, "%s", rccl_path()); }
  return 0;
}

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
  // Exchange forms no multi-rank run has ever executed (every box so far had one GPU; one-rank groups, where each collective
  // is a copy, are all that ran): refused for real groups unless the caller says it knows (DQNHIP_DP_UNVERIFIED_OK — what
  // tests/test_gpu_dp_native.py passes, the run that would verify them).  One-rank groups stay open: they pin the launches.
  if (h->cfg.dp_world > 1 && (flags & (DQNHIP_DP_PER_LAYER | DQNHIP_DP_SHARD_OPT)) && !(flags & DQNHIP_DP_UNVERIFIED_OK))
    return fail("dp_init: DQNHIP_DP_PER_LAYER / DQNHIP_DP_SHARD_OPT have never run on more than one rank (unverified on real links, and priced "
                "net-negative in DESIGN.md 6); the replicated one-bucket exchange is the supported form — pass DQNHIP_DP_UNVERIFIED_OK to try them");
  ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
  NCCLCHK(ncclCommInitRank(&h->comm, h->cfg.dp_world, uid, h->cfg.dp_rank));
  // Which RCCL is this?  libdqnhip.so asks for librccl.so.1; a process that already hosts PyTorch gets the build bundled with
  // torch (the loader matches the SONAME), a bare C++ / ctypes host gets /opt/rocm's.  Either is fine — a GROUP mixing two
  // builds is not: every rank contributes {version, -version} to a max all-reduce and all must see the same number.
  {
    int ver = 0;
    NCCLCHK(ncclGetVersion(&ver));
    h->rccl_version = ver;
    int* dv = nullptr;
    const int hv[2] = {ver, -ver};
    int got[2] = {0, 0};
    // every step checked, dv freed on every path: a failed copy must not read as "all ranks agree" ({0, 0} satisfies got[0] == -got[1])
    hipError_t e = hipMalloc(&dv, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(dv, hv, sizeof hv, hipMemcpyHostToDevice, h->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = ncclAllReduce(dv, dv, 2, ncclInt32, ncclMax, h->comm, h->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(got, dv, sizeof got, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(h->stream);
    if (dv) hipFree(dv);
    if (e != hipSuccess || r != ncclSuccess || got[0] != -got[1] || got[0] != ver) {
      ncclCommDestroy(h->comm); h->comm = nullptr;
      if (e != hipSuccess) return fail("dp_init: the version cross-check failed: %s", hipGetErrorString(e));
      if (r != ncclSuccess) return fail("dp_init: the version all-reduce failed: %s", ncclGetErrorString(r));
      return fail("dp_init: the ranks of this group loaded different RCCL builds (versions %d .. %d; this rank: %d from %s) — start every rank from "
                  "the same kind of host process (all with PyTorch's bundled librccl, or all with /opt/rocm's)", -got[1], got[0], ver, rccl_path());
    }
  }
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
  drop_graphs(h);
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
  h->epoch += 1;
  if (!h->comm) return fail("dp_broadcast_params: no communicator (call dqnhip_dp_init first)");
  if (root < 0 || root >= h->cfg.dp_world) return fail("bad root %d", root);
  HIPCHK(hipSetDevice(h->cfg.device));
  return dp_broadcast(h, root);
}

int dqnhip_dp_update(dqnhip_handle h, const int32_t* idx_host) {
  if (!h) return fail("null handle");
  h->epoch += 1;
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
  h->epoch += 1;
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
namespace dqnhip_host {
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
}  // extern "C"
