// dqn_dropin.cpp — what replaces the reference's src/dqn.cpp in its build: the out-of-line part
// of include/dqn.hpp's `dqn::DQN` (src/dqn.hpp:56-134) and the free functions (:204-242) as a thin
// host adaptor over the C-ABI of libdqnhip.so.  It uses the same glog / gflags / boost / caffe
// NAMES as the file it replaces, so it builds against the real libraries in the reference's
// environment and against include/shim/ where they are absent (this image).  All arithmetic is in
// the library; what stays here is what must stay on the host to keep the reference's behaviour:
// the std::mt19937 draws in the reference's call order (epsilon per SelectActions call :700,
// GetRandomActorOutput :664-682, SampleTransitionsFromMemory :501-509, SampleAction :180-194),
// the Update() gate / smoothed logs / snapshot cadence (:799-826), and the 11 learner gflags
// (:21-31), which must be DEFINED here because the driver's ParseCommandLineFlags owns the single
// flag namespace (src/dqn_main.cpp:393).
#include "dqn.hpp"

#include <glog/logging.h>
#include <gflags/gflags.h>

#include <chrono>
#include <cstdio>
#include <sstream>

namespace dqn {

using namespace hfo;

// the reference's learner flags, same names / defaults / help (src/dqn.cpp:21-31)
DEFINE_int32(seed, 0, "Seed the RNG. Default: time");
DEFINE_double(tau, .001, "Step size for soft updates.");
DEFINE_int32(soft_update_freq, 1, "Do SoftUpdateNet this frequently");
DEFINE_double(gamma, .99, "Discount factor of future rewards (0,1]");
DEFINE_int32(memory, 500000, "Capacity of replay memory");
DEFINE_int32(memory_threshold, 1000, "Number of transitions required to start learning");
DEFINE_int32(loss_display_iter, 1000, "Frequency of loss display");
DEFINE_int32(snapshot_freq, 10000, "Frequency (steps) snapshots");
DEFINE_bool(remove_old_snapshots, true, "Remove old snapshots when writing more recent ones.");
DEFINE_bool(snapshot_memory, true, "Snapshot the replay memory along with the network.");
DEFINE_double(beta, .5, "Mix between off-policy and on-policy updates.");
// MI355X learner flags (no counterpart in the reference; defaults reproduce its behaviour)
DEFINE_int32(minibatch, kMinibatchSize, "Minibatch size of the HIP learner (reference: kMinibatchSize = 32); multiple of 32.");
DEFINE_int32(select_actions_cap, 0, "Largest batch SelectActions accepts. 0: the reference's CHECK_LE(states_batch.size(), kMinibatchSize) "
                                    "(src/dqn.cpp:699) with this learner's -minibatch; -1: any batch (the device path has no MemoryData layer); n > 0: n.");
DEFINE_int32(hip_device, 0, "HIP device ordinal of this process's learners.");
DEFINE_bool(hip_graph, true, "Replay each update as one captured hipGraph.");
DEFINE_string(precision, "fp32", "fp32 (exact-fp32 MFMA, the parity path) or fp16 (fp16 MFMA operands, fp32 accumulate).");
DEFINE_bool(device_sampling, false, "Sample minibatch indices on the device (counter-based) instead of the host std::mt19937.");
DEFINE_bool(chained_updates, true, "UpdateActorCritic() tells the library which indices the NEXT call will draw (a copy of the engine runs ahead; "
            "dqnhip_update_chained): bursts of Update() get the launch schedule of a multi-update graph.  Same results, same RNG order.");
DEFINE_bool(pipelined_stats, false, "UpdateActorCritic() returns the (loss, avg_q) of the PREVIOUS update (dqnhip_update_pipelined): "
                                    "the device does not idle on the per-update read-back; the logged / smoothed values lag by one update.");
// Data parallelism for the UNCHANGED driver: start one process of this binary per GPU — each with its own HFO
// workers, its own replay shard, its own -save prefix and -hip_device — and the ranks' gradients are summed by RCCL inside
// libdqnhip.so twice per update (after the critic's backward and after the actor's).  Every rank's k-th UpdateActorCritic()
// pairs with every other rank's k-th: a rank that asks for an update early simply waits inside the collective until the
// others ask for theirs (episodes differ in length), and once max_iter is reached Update() becomes a no-op on every rank at
// the same update, so nobody is left waiting at the end.
DEFINE_int32(dp_world, 1, "Data-parallel group size (processes of this binary, one per GPU). 1: no group unless -dp_rendezvous is given.");
DEFINE_int32(dp_rank, 0, "This process's rank in the data-parallel group.");
DEFINE_string(dp_rendezvous, "", "File path shared by the ranks of the group (dqnhip_dp_init_file; needs no launcher). Empty: no data parallelism.");
DEFINE_bool(dp_half_grads, false, "Exchange gradients as bf16 (half the bytes on the links; meant for -precision fp16).");
DEFINE_int32(hip_agent_device_stride, 0, "Agent thread t of this process uses HIP device -hip_device + t * stride (0: all agents on -hip_device). "
                                         "BASELINE configs[3] on 4 GPUs = 2 processes x 2 agents, -dp_world 2, stride 2: agent a's rank r on GPU 2a + r.");


#define DQNHIP_CK(call) CHECK((call) == 0) << dqnhip_last_error()

namespace {

// tower widths = num_output of the InnerProduct layers named ip<i>_layer (src/dqn.cpp:406)
std::vector<int> TowerWidths(const caffe::NetParameter& np) {
  std::vector<int> widths;
  for (int i = 0; i < np.layer_size(); ++i) {
    const auto& l = np.layer(i);
    if (l.type() != "InnerProduct") continue;
    const std::string want = "ip" + std::to_string(widths.size() + 1) + "_layer";
    if (l.name() == want) widths.push_back(l.inner_product_param().num_output());
  }
  return widths;
}

void AddLayer(caffe::NetParameter& np, const std::string& name, const std::string& type,
              const std::vector<std::string>& bottoms, const std::vector<std::string>& tops) {
  caffe::LayerParameter* l = np.add_layer();
  l->set_name(name); l->set_type(type);
  for (const auto& b : bottoms) l->add_bottom(b);
  for (const auto& t : tops) l->add_top(t);
}
void AddMemoryData(caffe::NetParameter& np, const std::string& name, const std::vector<std::string>& tops, int n, int c, int h, int w) {
  AddLayer(np, name, "MemoryData", {}, tops);
  auto* p = np.mutable_layer(np.layer_size() - 1)->mutable_memory_data_param();
  p->set_batch_size(n); p->set_channels(c); p->set_height(h); p->set_width(w);
}
void AddInnerProduct(caffe::NetParameter& np, const std::string& name, const std::string& bottom, const std::string& top, int num_output) {
  AddLayer(np, name, "InnerProduct", {bottom}, {top});
  np.mutable_layer(np.layer_size() - 1)->mutable_inner_product_param()->set_num_output(num_output);   // gaussian(0.01) filler, :348-352
}
// ip<i>_layer (InnerProduct) + ip<i>_relu_layer (ReLU 0.01, in place) per width (src/dqn.cpp:400-416)
std::string AddTower(caffe::NetParameter& np, std::string input, const std::vector<int>& widths) {
  for (size_t i = 1; i <= widths.size(); ++i) {
    const std::string top = "ip" + std::to_string(i);
    AddInnerProduct(np, top + "_layer", input, top, widths[i - 1]);
    AddLayer(np, top + "_relu_layer", "ReLU", {top}, {top});
    np.mutable_layer(np.layer_size() - 1)->mutable_relu_param()->set_negative_slope(0.01f);
    input = top;
  }
  return input;
}
const std::vector<int> kDefaultTower = {1024, 512, 256, 128};     // src/dqn.cpp:425, 449

}  // namespace

caffe::NetParameter CreateActorNet(int state_size) {
  caffe::NetParameter np;
  np.set_name("Actor");
  np.set_force_backward(true);
  AddMemoryData(np, state_input_layer_name, {states_blob_name, "dummy1"}, kMinibatchSize, kStateInputCount, state_size, 1);
  AddLayer(np, "silence", "Silence", {"dummy1"}, {});
  const std::string top = AddTower(np, states_blob_name, kDefaultTower);
  AddInnerProduct(np, "action_layer", top, actions_blob_name, kActionSize);
  AddInnerProduct(np, "actionpara_layer", top, action_params_blob_name, kActionParamSize);
  return np;
}

caffe::NetParameter CreateCriticNet(int state_size) {
  caffe::NetParameter np;
  np.set_name("Critic");
  np.set_force_backward(true);
  AddMemoryData(np, state_input_layer_name, {states_blob_name, "dummy1"}, kMinibatchSize, kStateInputCount, state_size, 1);
  AddMemoryData(np, action_input_layer_name, {actions_blob_name, "dummy2"}, kMinibatchSize, kStateInputCount, kActionSize, 1);
  AddMemoryData(np, action_params_input_layer_name, {action_params_blob_name, "dummy3"}, kMinibatchSize, kStateInputCount, kActionParamSize, 1);
  AddMemoryData(np, target_input_layer_name, {targets_blob_name, "dummy4"}, kMinibatchSize, 1, 1, 1);
  AddLayer(np, "silence", "Silence", {"dummy1", "dummy2", "dummy3", "dummy4"}, {});
  AddLayer(np, "concat", "Concat", {states_blob_name, actions_blob_name, action_params_blob_name}, {"state_actions"});
  np.mutable_layer(np.layer_size() - 1)->mutable_concat_param()->set_axis(2);
  const std::string top = AddTower(np, "state_actions", kDefaultTower);
  AddInnerProduct(np, q_values_layer_name, top, q_values_blob_name, 1);
  AddLayer(np, "loss", "EuclideanLoss", {q_values_blob_name, targets_blob_name}, {loss_blob_name});
  return np;
}

int GetParamOffset(const action_t action, const int arg_num) {
  if (arg_num < 0 || arg_num > 1) return -1;
  switch (action) {
    case DASH: return arg_num;
    case TURN: return arg_num == 0 ? 2 : -1;
    case TACKLE: return arg_num == 0 ? 3 : -1;
    case KICK: return 4 + arg_num;
    default: LOG(FATAL) << "Unrecognized action: " << action;
  }
  return -1;
}

Action GetAction(const ActorOutput& actor_output) {
  ActorOutput logits(actor_output);
  logits[TACKLE] = -99999;                                    // TACKLE is never chosen (src/dqn.cpp:198)
  const action_t best = (action_t)std::distance(logits.begin(), std::max_element(logits.begin(), logits.begin() + kActionSize));
  const int o1 = GetParamOffset(best, 0), o2 = GetParamOffset(best, 1);
  CHECK_GE(o1, 0);
  Action a;
  a.action = best;
  a.arg1 = actor_output[kActionSize + o1];
  a.arg2 = o2 < 0 ? 0 : actor_output[kActionSize + o2];
  return a;
}

std::string PrintActorOutput(const ActorOutput& o) {
  auto s = [](float v) { return std::to_string(v); };
  return "Dash(" + s(o[4]) + ", " + s(o[5]) + ")=" + s(o[0]) + ", Turn(" + s(o[6]) + ")=" + s(o[1]) + ", Tackle(" + s(o[7]) + ")=" + s(o[2]) +
         ", Kick(" + s(o[8]) + ", " + s(o[9]) + ")=" + s(o[3]);
}

std::vector<std::string> FilesMatchingRegexp(const std::string& regexp) {
  std::vector<char> buf(1 << 16);
  int32_t n = 0;
  while (dqnhip_files_matching_regexp(regexp.c_str(), buf.data(), buf.size(), &n) != 0) {
    CHECK(std::string(dqnhip_last_error()).find("buffer too small") != std::string::npos && buf.size() < (1u << 28)) << dqnhip_last_error();
    buf.resize(buf.size() * 4);
  }
  std::vector<std::string> out;
  std::istringstream ss(buf.data());
  for (std::string line; std::getline(ss, line);) if (!line.empty()) out.push_back(line);
  return out;
}
void RemoveFilesMatchingRegexp(const std::string& regexp) { DQNHIP_CK(dqnhip_remove_files_matching_regexp(regexp.c_str())); }
void RemoveSnapshots(const std::string& regexp, int min_iter) { DQNHIP_CK(dqnhip_remove_snapshots(regexp.c_str(), min_iter)); }
void FindLatestSnapshot(const std::string& snapshot_prefix, std::string& actor_snapshot, std::string& critic_snapshot,
                        std::string& memory_snapshot) {
  char a[4096], c[4096], m[4096];
  DQNHIP_CK(dqnhip_find_latest_snapshot(snapshot_prefix.c_str(), a, c, m, sizeof a));
  if (a[0]) actor_snapshot = a;
  if (c[0]) critic_snapshot = c;
  if (m[0]) memory_snapshot = m;
}
int FindHiScore(const std::string& snapshot_prefix) {
  int32_t s = 0;
  DQNHIP_CK(dqnhip_find_hiscore(snapshot_prefix.c_str(), &s));
  return s;
}

// ---------------------------------------------------------------------------------------------
DQN::DQN(caffe::SolverParameter& actor_solver_param, caffe::SolverParameter& critic_solver_param, std::string save_path,
         int state_size, int tid)
    : actor_solver_param_(actor_solver_param), critic_solver_param_(critic_solver_param), replay_memory_capacity_(FLAGS_memory),
      gamma_(FLAGS_gamma), random_engine(), smoothed_critic_loss_(0), smoothed_actor_loss_(0), last_snapshot_iter_(0),
      save_path_(save_path), state_size_(state_size), tid_(tid), unum_(0), minibatch_(FLAGS_minibatch), h_(nullptr) {
  unsigned seed;
  if (FLAGS_seed <= 0) {
    seed = (unsigned)std::chrono::system_clock::now().time_since_epoch().count();
    LOG(INFO) << "Seeding RNG to time (seed = " << seed << ")";
  } else {
    seed = (unsigned)FLAGS_seed;
    LOG(INFO) << "Seeding RNG with seed = " << FLAGS_seed;
  }
  random_engine.seed(seed);
  // -gpu=false selects Caffe's CPU solver in the reference (KeepPlayingGames: Caffe::set_mode(Caffe::CPU),
  // src/dqn_main.cpp:208-212).  This library replaces the GPU path only and by design has no CPU fallback — a learner that
  // silently computed somewhere else would void every measurement — so the flag ends here, through the driver's own logging
  // path, with what to do instead of a bare CHECK on an internal condition.
  if (caffe::Caffe::mode() != caffe::Caffe::GPU)
    LOG(FATAL) << "[Agent" << tid << "] -gpu=false: Caffe CPU mode (src/dqn_main.cpp:208-212) is not provided by the MI355X drop-in "
               << "(libdqnhip.so has no CPU backend). Run the reference's own Caffe build for the CPU solver, or start this binary with -gpu=true.";
  std::vector<int> wa = TowerWidths(actor_solver_param_.net_param()), wc = TowerWidths(critic_solver_param_.net_param());
  if (wa.empty()) wa = kDefaultTower;
  if (wc.empty()) wc = kDefaultTower;
  CHECK(wa == wc) << "actor and critic towers must have the same widths";
  CHECK_LE(wa.size(), (size_t)DQNHIP_MAX_HIDDEN);
  dqnhip_config c;
  dqnhip_default_config(&c, state_size);
  c.minibatch = FLAGS_minibatch;
  c.num_hidden = (int)wa.size();
  for (size_t i = 0; i < wa.size(); ++i) c.hidden[i] = wa[i];
  c.replay_capacity = FLAGS_memory;
  c.soft_update_freq = FLAGS_soft_update_freq;
  c.gamma = FLAGS_gamma; c.beta = FLAGS_beta; c.tau = FLAGS_tau;
  // each solver's own hyper-parameters; the fused optimiser pass takes ONE (momentum, momentum2,
  // delta, clip) pair, so the two solvers must agree on those (the driver sets them from the same flags)
  c.actor_lr = actor_solver_param_.base_lr(); c.critic_lr = critic_solver_param_.base_lr();
  CHECK(actor_solver_param_.momentum() == critic_solver_param_.momentum() && actor_solver_param_.momentum2() == critic_solver_param_.momentum2() &&
        actor_solver_param_.clip_gradients() == critic_solver_param_.clip_gradients() && actor_solver_param_.delta() == critic_solver_param_.delta())
      << "actor and critic solvers must share momentum / momentum2 / delta / clip_gradients";
  c.momentum = critic_solver_param_.momentum(); c.momentum2 = critic_solver_param_.momentum2();
  c.delta = critic_solver_param_.delta(); c.clip_gradients = critic_solver_param_.clip_gradients();
  // -lr_policy (src/dqn_main.cpp:36, 255-256): the fused optimiser pass applies base_lr as it stands, i.e. Caffe's "fixed" policy
  // (the reference's default); anything else would silently train with a different schedule than the user asked for
  for (const caffe::SolverParameter* sp : {&actor_solver_param_, &critic_solver_param_})
    CHECK(sp->lr_policy().empty() || sp->lr_policy() == "fixed") << "only -lr_policy fixed is implemented (got '" << sp->lr_policy() << "')";
  CHECK(actor_solver_param_.type() == "Adam" && critic_solver_param_.type() == "Adam") << "only the Adam solver is implemented (-solver Adam, the reference's default)";
  c.device = FLAGS_hip_device + tid * FLAGS_hip_agent_device_stride;
  c.use_graph = FLAGS_hip_graph ? 1 : 0;
  c.seed = seed;
  CHECK(FLAGS_precision == "fp32" || FLAGS_precision == "fp16") << "-precision must be fp32 or fp16";
  c.precision = FLAGS_precision == "fp16" ? DQNHIP_FP16 : DQNHIP_FP32;
  dp_ = !FLAGS_dp_rendezvous.empty();
  CHECK(dp_ || FLAGS_dp_world == 1) << "-dp_world > 1 needs -dp_rendezvous <path shared by the ranks>";
  if (dp_) {
    CHECK(!FLAGS_pipelined_stats) << "-pipelined_stats is a single-learner option (the data-parallel update reports its own scalars)";
    // Every agent thread's learner is its own group with its own communicator.  Two communicators whose collectives are
    // enqueued on ONE device in an order nothing coordinates is what RCCL documents as unsafe, and with a shared replay the
    // driver's global MTX around every Update() burst (src/dqn_main.cpp:358-362) closes a cross-process cycle (rank 0's agent 0
    // holds MTX inside a collective whose peer waits for rank 1's MTX, held by agent 1, whose peer waits for rank 0's MTX).
    // So: a second agent needs its own device, and ShareReplayMemory refuses data-parallel learners (below).
    CHECK(tid == 0 || FLAGS_hip_agent_device_stride > 0)
        << "-dp_rendezvous with more than one agent thread needs -hip_agent_device_stride > 0 (one device per agent: each agent's "
        << "data-parallel group has its own RCCL communicator, and two communicators must not share a device)";
    c.dp_world = FLAGS_dp_world; c.dp_rank = FLAGS_dp_rank;
  }
  DQNHIP_CK(dqnhip_create(&c, &h_));
  if (dp_) {
    // one communicator per agent thread's learner: agents are independent DQNs (src/dqn_main.cpp:264), each its own group
    const std::string rv = FLAGS_dp_rendezvous + "_agent" + std::to_string(tid);
    LOG(INFO) << "[Agent" << tid << "] data-parallel rank " << FLAGS_dp_rank << " of " << FLAGS_dp_world << " (rendezvous " << rv << ")";
    DQNHIP_CK(dqnhip_dp_init_file(h_, rv.c_str(), FLAGS_dp_half_grads ? DQNHIP_DP_HALF_GRADS : 0, 300));
    // every rank re-synchronises at its first Update(), whether or not IT restored anything: a rank that found a snapshot under
    // its own -save prefix and a rank that did not must still enter the same collective (see SyncReplicasIfPending)
    dp_sync_pending_ = true;
    int32_t ver = 0; char path[1024] = {0};
    if (dqnhip_dp_info(&ver, path, sizeof path) == 0) LOG(INFO) << "[Agent" << tid << "] RCCL " << ver << " from " << path;
  }
}

// The driver restores / preloads AFTER construction (RestoreActorSolver, RestoreCriticSolver, LoadActorWeights, LoadCriticWeights:
// src/dqn_main.cpp:268-282), i.e. after dqnhip_dp_init's broadcast.  Ranks that resume from their own -save prefixes would then
// train diverged replicas on summed gradients, and ranks whose iteration counters differ would leave the max_iter gate of
// Update() at different updates — one of them waiting in a collective for ever.  So every such call re-arms a broadcast of
// rank 0's weights, Adam history and iterations, taken at the next Update() / UpdateActorCritic(): the first point every rank
// passes in the same order.  The flag is also set by the constructor, so the FIRST update of every rank broadcasts whether or not
// that rank restored anything (symmetric: one rank with a snapshot and one without still meet in the same collective).
// That broadcast is symmetric only BEFORE the first update: afterwards a Restore* / Load* on one rank alone would arm it on that rank
// alone (a collective with no peer), and ranks may pass Update() a different number of times per episode.  So after the group's
// first update these four calls are refused on a data-parallel learner (RearmReplicaSync): restore before training starts, on every
// rank alike — what the reference's driver does (src/dqn_main.cpp:268-282).
void DQN::SyncReplicasIfPending() {
  if (!dp_ || !dp_sync_pending_) return;
  LOG(INFO) << "[Agent" << tid_ << "] data-parallel: re-synchronising the replicas from rank 0 (weights, Adam history, iterations)";
  DQNHIP_CK(dqnhip_dp_broadcast_params(h_, 0));
  dp_sync_pending_ = false;
  dp_synced_once_ = true;
  last_snapshot_iter_ = max_iter();
}
void DQN::RearmReplicaSync(const char* what) {
  if (!dp_) return;
  if (dp_synced_once_)
    LOG(FATAL) << "[Agent" << tid_ << "] " << what << " on a data-parallel learner after its first Update(): the re-synchronising broadcast is a "
               << "collective that only this rank would enter.  Restore / load before the first update, with the same calls on every rank.";
  dp_sync_pending_ = true;
}

DQN::~DQN() { DQNHIP_CK(dqnhip_destroy(h_)); }

// src/dqn.cpp:487-498: a wall-clock timer around `iterations` calls of UpdateActorCritic() — host-drawn
// indices and a blocking (loss, avg_q) per call, unless -device_sampling / -pipelined_stats say otherwise
void DQN::Benchmark(int iterations) {
  LOG(INFO) << "*** Benchmark begins ***";
  const auto t0 = std::chrono::steady_clock::now();
  if (FLAGS_device_sampling && !FLAGS_pipelined_stats && !dp_ && iterations > 0) {
    // nothing between the updates needs the host: the whole loop is one call (sixteen updates per hipGraph launch)
    float loss = 0.0f, avg_q = 0.0f;
    DQNHIP_CK(dqnhip_update_async_n(h_, iterations));
    DQNHIP_CK(dqnhip_read_stats(h_, &loss, &avg_q));          // blocks until the last update is done
  } else
    for (int i = 0; i < iterations; ++i) UpdateActorCritic();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  LOG(INFO) << "Average Update: " << ms / iterations << " ms.";
  LOG(INFO) << "*** Benchmark ends ***";
}

void DQN::RestoreActorSolver(const std::string& f) { RearmReplicaSync("RestoreActorSolver"); DQNHIP_CK(dqnhip_solver_restore(h_, DQNHIP_ACTOR, f.c_str())); last_snapshot_iter_ = max_iter(); }
void DQN::RestoreCriticSolver(const std::string& f) { RearmReplicaSync("RestoreCriticSolver"); DQNHIP_CK(dqnhip_solver_restore(h_, DQNHIP_CRITIC, f.c_str())); last_snapshot_iter_ = max_iter(); }
void DQN::LoadActorWeights(const std::string& f) { RearmReplicaSync("LoadActorWeights"); DQNHIP_CK(dqnhip_load_caffemodel(h_, DQNHIP_ACTOR, f.c_str())); }
void DQN::LoadCriticWeights(const std::string& f) { RearmReplicaSync("LoadCriticWeights"); DQNHIP_CK(dqnhip_load_caffemodel(h_, DQNHIP_CRITIC, f.c_str())); }
void DQN::LoadReplayMemory(const std::string& f) {
  LOG(INFO) << "Loading replay memory from " << f;
  DQNHIP_CK(dqnhip_load_replay_memory(h_, f.c_str()));
  LOG(INFO) << "replay_mem_size = " << memory_size();
}
void DQN::SnapshotReplayMemory(const std::string& f) { DQNHIP_CK(dqnhip_snapshot_replay_memory(h_, f.c_str())); }

void DQN::Snapshot() { Snapshot(save_path_, FLAGS_remove_old_snapshots, FLAGS_snapshot_memory); }
void DQN::Snapshot(const std::string& snapshot_prefix, bool remove_old, bool snapshot_memory) {
  if (snapshot_memory) LOG(INFO) << "Snapshotting memory to " << snapshot_prefix << "_iter_" << max_iter() << ".replaymemory";
  DQNHIP_CK(dqnhip_snapshot(h_, save_path_.c_str(), snapshot_prefix.c_str(), remove_old, snapshot_memory));
  LOG(INFO) << "Snapshotting Finished!";
}

ActorOutput DQN::GetRandomActorOutput() {
  auto u = [this](float lo, float hi) { return std::uniform_real_distribution<float>(lo, hi)(random_engine); };
  ActorOutput o;
  for (int i = 0; i < kActionSize; ++i) o[i] = u(-1.0, 1.0);
  o[kActionSize + 0] = u(-100.0, 100.0);      // Dash Power   (draw order = src/dqn.cpp:669-680)
  o[kActionSize + 1] = u(-180.0, 180.0);      // Dash Angle
  o[kActionSize + 2] = u(-180.0, 180.0);      // Turn Angle
  o[kActionSize + 3] = u(-180.0, 180.0);      // Tackle Angle
  o[kActionSize + 4] = u(0.0, 100.0);         // Kick Power
  o[kActionSize + 5] = u(-180.0, 180.0);      // Kick Angle
  return o;
}

ActorOutput DQN::SelectAction(const InputStates& last_states, const double epsilon) {
  return SelectActions(std::vector<InputStates>{{last_states}}, epsilon)[0];
}

std::vector<ActorOutput> DQN::SelectActions(const std::vector<InputStates>& states_batch, const double epsilon) {
  CHECK(epsilon >= 0.0 && epsilon <= 1.0);
  // :699 CHECK_LE(states_batch.size(), kMinibatchSize): the MemoryData layer is that wide there; here the cap is kept as the
  // reference's behaviour with this learner's minibatch, and -select_actions_cap widens it (the device path takes any n)
  const int cap = FLAGS_select_actions_cap == 0 ? minibatch_ : FLAGS_select_actions_cap;
  if (cap >= 0) CHECK_LE((int)states_batch.size(), cap);
  std::vector<ActorOutput> out(states_batch.size());
  if (std::uniform_real_distribution<double>(0.0, 1.0)(random_engine) < epsilon) {      // ONE draw per call (:700)
    for (auto& o : out) o = GetRandomActorOutput();
    return out;
  }
  if (states_batch.empty()) return out;
  std::vector<float> s(states_batch.size() * (size_t)state_size_);
  for (size_t n = 0; n < states_batch.size(); ++n) {
    const StateDataSp& sp = states_batch[n][0];
    CHECK(sp) << "null state";
    CHECK_EQ((int)sp->size(), state_size_);
    std::copy(sp->begin(), sp->end(), s.begin() + n * state_size_);
  }
  DQNHIP_CK(dqnhip_select_actions(h_, s.data(), (int)states_batch.size(), out[0].data()));
  return out;
}

Action DQN::SampleAction(const ActorOutput& actor_output) {
  const float dash = std::max(0., actor_output[DASH] + 1.0), turn = std::max(0., actor_output[TURN] + 1.0);
  const float kick = std::max(0., actor_output[KICK] + 1.0);
  std::discrete_distribution<int> dist{dash, turn, 0 /* tackle removed */, kick};
  const action_t act = (action_t)dist(random_engine);
  const int o1 = GetParamOffset(act, 0), o2 = GetParamOffset(act, 1);
  CHECK_GE(o1, 0);
  Action a;
  a.action = act;
  a.arg1 = actor_output[kActionSize + o1];
  a.arg2 = o2 < 0 ? 0 : actor_output[kActionSize + o2];
  return a;
}

float DQN::EvaluateAction(const InputStates& input_states, const ActorOutput& action) {
  float q = 0;
  DQNHIP_CK(dqnhip_critic_forward(h_, DQNHIP_CRITIC, input_states[0]->data(), action.data(), 1, &q));
  return q;
}

void DQN::AddTransition(const Transition& t) {
  const auto& next = std::get<4>(t);
  DQNHIP_CK(dqnhip_add_transition(h_, std::get<0>(t)[0]->data(), std::get<1>(t).data(), std::get<2>(t), std::get<3>(t),
                                  next ? (*next)->data() : nullptr, next ? 0 : 1));
}

void DQN::AddTransitions(const std::vector<Transition>& ts) {
  const size_t n = ts.size(), S = (size_t)state_size_;
  if (n == 0) { DQNHIP_CK(dqnhip_add_transitions(h_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0)); return; }   // (:776 still evicts one from a full deque)
  std::vector<float> s(n * S), nx(n * S, 0.f), a(n * (kActionSize + kActionParamSize)), r(n), mc(n);
  std::vector<uint8_t> term(n);
  for (size_t i = 0; i < n; ++i) {
    const Transition& t = ts[i];
    std::copy(std::get<0>(t)[0]->begin(), std::get<0>(t)[0]->end(), s.begin() + i * S);
    std::copy(std::get<1>(t).begin(), std::get<1>(t).end(), a.begin() + i * (kActionSize + kActionParamSize));
    r[i] = std::get<2>(t); mc[i] = std::get<3>(t);
    const auto& next = std::get<4>(t);
    term[i] = next ? 0 : 1;                                  // terminal <=> next_state is none (:878)
    if (next) std::copy((*next)->begin(), (*next)->end(), nx.begin() + i * S);
  }
  DQNHIP_CK(dqnhip_add_transitions(h_, s.data(), a.data(), r.data(), mc.data(), nx.data(), term.data(), (int)n));
}

void DQN::LabelTransitions(std::vector<Transition>& ts) {
  CHECK_GT(ts.size(), 0u) << "Need at least one transition to label.";
  std::vector<float> r(ts.size()), mc(ts.size());
  for (size_t i = 0; i < ts.size(); ++i) r[i] = std::get<2>(ts[i]);
  DQNHIP_CK(dqnhip_label_transitions(gamma_, r.data(), (int)ts.size(), mc.data()));
  for (size_t i = 0; i < ts.size(); ++i) std::get<3>(ts[i]) = mc[i];
}

void DQN::Update() {
  SyncReplicasIfPending();       // (before the gates below: they read the iteration counters the broadcast carries)
  // data parallel: every rank's iteration counters advance together, so every rank stops updating at the same update — the
  // driver's own loops only look at max_iter() between bursts of Update() calls (src/dqn_main.cpp:354-362), and a rank left
  // alone in a collective would wait for ever
  if (dp_ && actor_solver_param_.max_iter() > 0 && max_iter() >= actor_solver_param_.max_iter()) return;
  if (memory_size() < FLAGS_memory_threshold) return;
  const std::pair<float, float> res = UpdateActorCritic();
  if (critic_iter() % FLAGS_loss_display_iter == 0) {
    LOG(INFO) << "[Agent" << tid_ << "] Critic Iteration " << critic_iter() << ", loss = " << smoothed_critic_loss_;
    smoothed_critic_loss_ = 0;
  }
  smoothed_critic_loss_ += res.first / float(FLAGS_loss_display_iter);
  if (actor_iter() % FLAGS_loss_display_iter == 0) {
    LOG(INFO) << "[Agent" << tid_ << "] Actor Iteration " << actor_iter() << ", avg_q_value = " << smoothed_actor_loss_;
    smoothed_actor_loss_ = 0;
  }
  smoothed_actor_loss_ += res.second / float(FLAGS_loss_display_iter);
  if (critic_iter() >= last_snapshot_iter_ + FLAGS_snapshot_freq || actor_iter() >= last_snapshot_iter_ + FLAGS_snapshot_freq) {
    Snapshot();
    last_snapshot_iter_ = max_iter();
  }
}

std::vector<int> DQN::SampleTransitionsFromMemory(int n) {
  std::vector<int> idx(n);
  const int size = memory_size();
  for (int& i : idx) i = std::uniform_int_distribution<int>(0, size - 1)(random_engine);
  return idx;
}

std::vector<InputStates> DQN::SampleStatesFromMemory(int n) {
  const std::vector<int> idx = SampleTransitionsFromMemory(n);
  std::vector<float> flat((size_t)n * state_size_);
  DQNHIP_CK(dqnhip_sample_states(h_, idx.data(), n, flat.data()));
  std::vector<InputStates> out(n);
  for (int i = 0; i < n; ++i)
    out[i][0] = std::make_shared<StateData>(flat.begin() + (size_t)i * state_size_, flat.begin() + (size_t)(i + 1) * state_size_);
  return out;
}

std::pair<float, float> DQN::UpdateActorCritic() {
  SyncReplicasIfPending();
  if (FLAGS_device_sampling) {
    float loss = 0, avgq = 0;
    if (dp_) { DQNHIP_CK(dqnhip_dp_update(h_, nullptr)); DQNHIP_CK(dqnhip_read_stats(h_, &loss, &avgq)); }
    else DQNHIP_CK(dqnhip_update(h_, nullptr, &loss, &avgq));                 // CHECK(isfinite(target / loss)): the call fails
    return std::make_pair(loss, avgq);
  }
  if (dp_ || FLAGS_pipelined_stats || !FLAGS_chained_updates) return UpdateActorCritic(SampleTransitionsFromMemory(minibatch_));
  // The driver calls this in bursts (src/dqn_main.cpp:359-361), each call drawing its indices from random_engine (:501-509).  What the
  // NEXT call will draw is known now — unless something else draws from the engine or the memory changes in between: take it from a
  // COPY of the engine and hand it to the library as a prediction (dqnhip_update_chained: the next update's gather and first layers
  // ride in this update's optimiser launches).  The next call adopts the copy's state only if the engine and the memory size are
  // exactly as the prediction left them; otherwise it draws afresh.  Either way random_engine advances as the reference's does.
  const int size = memory_size();
  std::vector<int> idx;
  if (spec_valid_ && size == spec_size_ && random_engine == spec_before_) { idx.swap(spec_idx_); random_engine = spec_after_; }
  else idx = SampleTransitionsFromMemory(minibatch_);
  spec_before_ = random_engine;
  spec_after_ = random_engine;
  spec_idx_.resize(minibatch_);
  for (int& i : spec_idx_) i = std::uniform_int_distribution<int>(0, size - 1)(spec_after_);
  spec_size_ = size; spec_valid_ = true;
  float loss = 0, avgq = 0;
  DQNHIP_CK(dqnhip_update_chained(h_, reinterpret_cast<const int32_t*>(idx.data()), reinterpret_cast<const int32_t*>(spec_idx_.data()), &loss, &avgq));
  return std::make_pair(loss, avgq);
}

std::pair<float, float> DQN::UpdateActorCritic(const std::vector<int>& transitions) {
  CHECK_EQ((int)transitions.size(), minibatch_);
  SyncReplicasIfPending();
  float loss = 0, avgq = 0;
  static_assert(sizeof(int) == sizeof(int32_t), "indices travel as int32");
  if (dp_) {   // this rank's rows of the global minibatch: phase 0 / all-reduce / phase 1 / all-reduce / phase 2 inside the library
    DQNHIP_CK(dqnhip_dp_update(h_, reinterpret_cast<const int32_t*>(transitions.data())));
    DQNHIP_CK(dqnhip_read_stats(h_, &loss, &avgq));
  } else if (FLAGS_pipelined_stats) DQNHIP_CK(dqnhip_update_pipelined(h_, reinterpret_cast<const int32_t*>(transitions.data()), &loss, &avgq));
  else DQNHIP_CK(dqnhip_update(h_, reinterpret_cast<const int32_t*>(transitions.data()), &loss, &avgq));
  return std::make_pair(loss, avgq);
}

void DQN::ClearReplayMemory() { DQNHIP_CK(dqnhip_clear_memory(h_)); }
int DQN::memory_size() const { int32_t n = 0; DQNHIP_CK(dqnhip_memory_size(h_, &n)); return n; }
int DQN::critic_iter() const { int32_t a = 0, c = 0; DQNHIP_CK(dqnhip_get_iters(h_, &a, &c)); return c; }
int DQN::actor_iter() const { int32_t a = 0, c = 0; DQNHIP_CK(dqnhip_get_iters(h_, &a, &c)); return a; }

void DQN::ShareParameters(DQN& other, int num_actor_layers_to_share, int num_critic_layers_to_share) {
  DQNHIP_CK(dqnhip_share_parameters(h_, other.h_, num_actor_layers_to_share, num_critic_layers_to_share));
}
// src/dqn.cpp:1037-1045 shares the blobs of two caffe::Layer objects; the nets here are parameter arenas
// in HBM without Layer objects, and the one caller (ShareParameters) is implemented on arena prefixes
void DQN::ShareLayer(caffe::Layer<float>&, caffe::Layer<float>&) {
  LOG(FATAL) << "DQN::ShareLayer: no caffe::Layer objects exist behind this DQN; use ShareParameters(other, n_actor, n_critic)";
}
void DQN::ShareReplayMemory(DQN& other) {
  // (see the constructor: a shared replay puts the driver's global MTX around collectives of two independent groups)
  CHECK(!dp_ && !other.dp_) << "ShareReplayMemory cannot be combined with -dp_rendezvous: the driver holds one mutex around every agent's "
                            << "Update() burst (src/dqn_main.cpp:358-362) and each agent's updates are collectives of its own group";
  DQNHIP_CK(dqnhip_share_replay_memory(h_, other.h_));
}

}  // namespace dqn
