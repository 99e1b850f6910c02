// dqn_adaptor.hpp — the reference's `dqn::DQN` class surface (src/dqn.hpp:16-244) as a thin,
// header-only C++ adaptor over the C-ABI of include/dqnhip.h.  This is what replaces
// src/dqn.cpp in the reference build: same type aliases, same method names and argument
// meaning, same abort-on-error convention, std::mt19937 epsilon / sampling draws kept on the
// host in the reference's call order (src/dqn.cpp:501-509, 664-711).
//
// Differences that are forced by the missing third-party headers in this image (and are one
// typedef away in the reference build): boost::optional -> std::optional, caffe::
// SolverParameter -> dqn::SolverParams (the fields src/dqn_main.cpp:247-262 sets), glog
// LOG(FATAL) -> DQN_FATAL (prints and aborts).  The learner gflags of src/dqn.cpp:21-31 are
// members of dqn::Flags.
#ifndef DQN_ADAPTOR_HPP_
#define DQN_ADAPTOR_HPP_

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <optional>
#include <random>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/dqnhip.h"

#define DQN_FATAL(...) do { std::fprintf(stderr, "F dqn: " __VA_ARGS__); std::fprintf(stderr, "\n"); std::abort(); } while (0)
#define DQN_CK(call) do { if ((call) != 0) DQN_FATAL("%s: %s", #call, dqnhip_last_error()); } while (0)

// src/hfo_game.hpp:7-11 (hfo::action_t values 0..3 are pinned by src/dqn.cpp:181-186)
enum action_t { DASH = 0, TURN = 1, TACKLE = 2, KICK = 3 };
struct Action { action_t action; float arg1; float arg2; };

namespace dqn {

constexpr auto kStateInputCount = 1;       // src/dqn.hpp:18
constexpr auto kActionSize = 4;            // :20
constexpr auto kActionParamSize = 6;       // :21

using ActorOutput = std::array<float, kActionSize + kActionParamSize>;
using StateData = std::vector<float>;
using StateDataSp = std::shared_ptr<StateData>;
using InputStates = std::array<StateDataSp, kStateInputCount>;
using Transition = std::tuple<InputStates, ActorOutput, float, float, std::optional<StateDataSp>>;

// the learner flags of src/dqn.cpp:21-31 (defaults are the reference's)
struct Flags {
  int seed = 0; double tau = .001; int soft_update_freq = 1; double gamma = .99;
  int memory = 500000; int memory_threshold = 1000; int loss_display_iter = 1000;
  int snapshot_freq = 10000; bool remove_old_snapshots = true; bool snapshot_memory = true;
  double beta = .5;
  int minibatch = 32;                                   // kMinibatchSize, src/dqn.hpp:19
  std::vector<int> hidden = {1024, 512, 256, 128};      // Tower sizes, src/dqn.cpp:425,449
};
// the solver fields src/dqn_main.cpp:249-262 sets
struct SolverParams { float base_lr = 1e-3f; float momentum = .95f; float momentum2 = .999f; float clip_gradients = 10.f; };

inline int GetParamOffset(const action_t action, const int arg_num = 0) {   // src/dqn.cpp:162-178
  if (arg_num < 0 || arg_num > 1) return -1;
  switch (action) {
    case DASH: return arg_num;
    case TURN: return arg_num == 0 ? 2 : -1;
    case TACKLE: return arg_num == 0 ? 3 : -1;
    case KICK: return 4 + arg_num;
  }
  DQN_FATAL("Unrecognized action: %d", (int)action);
}

inline Action GetAction(const ActorOutput& actor_output) {                  // src/dqn.cpp:196-208
  ActorOutput copy(actor_output);
  copy[TACKLE] = -99999;
  action_t max_act = (action_t)std::distance(copy.begin(), std::max_element(copy.begin(), copy.begin() + kActionSize));
  Action action;
  action.action = max_act;
  action.arg1 = actor_output[kActionSize + GetParamOffset(max_act, 0)];
  const int arg2_offset = GetParamOffset(max_act, 1);
  action.arg2 = arg2_offset < 0 ? 0 : actor_output[kActionSize + arg2_offset];
  return action;
}

// free functions of src/dqn.hpp:204-242 -------------------------------------------------------
inline void RemoveFilesMatchingRegexp(const std::string& regexp) { DQN_CK(dqnhip_remove_files_matching_regexp(regexp.c_str())); }
inline void FindLatestSnapshot(const std::string& snapshot_prefix, std::string& actor_snapshot,
                               std::string& critic_snapshot, std::string& memory_snapshot) {
  char a[4096], c[4096], m[4096];
  DQN_CK(dqnhip_find_latest_snapshot(snapshot_prefix.c_str(), a, c, m, sizeof a));
  if (a[0]) actor_snapshot = a;
  if (c[0]) critic_snapshot = c;
  if (m[0]) memory_snapshot = m;
}
inline int FindHiScore(const std::string& snapshot_prefix) { int32_t s; DQN_CK(dqnhip_find_hiscore(snapshot_prefix.c_str(), &s)); return s; }
inline std::string PrintActorOutput(const ActorOutput& o) {        // src/dqn.cpp:210-216
  return "Dash(" + std::to_string(o[4]) + ", " + std::to_string(o[5]) + ")=" + std::to_string(o[0]) + ", Turn(" +
         std::to_string(o[6]) + ")=" + std::to_string(o[1]) + ", Tackle(" + std::to_string(o[7]) + ")=" + std::to_string(o[2]) +
         ", Kick(" + std::to_string(o[8]) + ", " + std::to_string(o[9]) + ")=" + std::to_string(o[3]);
}

class DQN {
 public:
  DQN(const SolverParams& actor_solver_param, const SolverParams& critic_solver_param, std::string save_path,
      int state_size, int tid, const Flags& flags = Flags())
      : flags_(flags), gamma_(flags.gamma), save_path_(std::move(save_path)), state_size_(state_size), tid_(tid) {
    dqnhip_config c;
    dqnhip_default_config(&c, state_size);
    c.minibatch = flags.minibatch;
    c.num_hidden = (int)flags.hidden.size();
    for (size_t i = 0; i < flags.hidden.size(); ++i) c.hidden[i] = flags.hidden[i];
    c.replay_capacity = flags.memory; c.soft_update_freq = flags.soft_update_freq;
    c.gamma = flags.gamma; c.beta = flags.beta; c.tau = flags.tau;
    c.actor_lr = actor_solver_param.base_lr; c.critic_lr = critic_solver_param.base_lr;
    c.momentum = critic_solver_param.momentum; c.momentum2 = critic_solver_param.momentum2;
    c.clip_gradients = critic_solver_param.clip_gradients;
    unsigned seed = flags.seed > 0 ? (unsigned)flags.seed                      // src/dqn.cpp:474-481
                                   : (unsigned)std::chrono::system_clock::now().time_since_epoch().count();
    random_engine.seed(seed);
    c.seed = seed;
    DQN_CK(dqnhip_create(&c, &h_));
  }
  ~DQN() { dqnhip_destroy(h_); }
  DQN(const DQN&) = delete;
  DQN& operator=(const DQN&) = delete;

  void Benchmark(int iterations = 1000) {                                     // src/dqn.cpp:487-498
    float ms = 0;
    DQN_CK(dqnhip_benchmark(h_, 0, iterations, &ms));
    std::printf("Average Update: %g ms.\n", ms);
  }

  ActorOutput GetRandomActorOutput() {                                        // src/dqn.cpp:664-682
    ActorOutput o;
    for (int i = 0; i < kActionSize; ++i) o[i] = std::uniform_real_distribution<float>(-1.0, 1.0)(random_engine);
    o[kActionSize + 0] = std::uniform_real_distribution<float>(-100.0, 100.0)(random_engine);
    o[kActionSize + 1] = std::uniform_real_distribution<float>(-180.0, 180.0)(random_engine);
    o[kActionSize + 2] = std::uniform_real_distribution<float>(-180.0, 180.0)(random_engine);
    o[kActionSize + 3] = std::uniform_real_distribution<float>(-180.0, 180.0)(random_engine);
    o[kActionSize + 4] = std::uniform_real_distribution<float>(0.0, 100.0)(random_engine);
    o[kActionSize + 5] = std::uniform_real_distribution<float>(-180.0, 180.0)(random_engine);
    return o;
  }

  ActorOutput SelectAction(const InputStates& input_states, double epsilon) { // src/dqn.cpp:684-686
    return SelectActions(std::vector<InputStates>{{input_states}}, epsilon)[0];
  }

  std::vector<ActorOutput> SelectActions(const std::vector<InputStates>& states_batch, double epsilon) {  // :695-711
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) DQN_FATAL("Check failed: epsilon >= 0.0 && epsilon <= 1.0");
    std::vector<ActorOutput> out(states_batch.size());
    if (std::uniform_real_distribution<double>(0.0, 1.0)(random_engine) < epsilon) {   // ONE draw per call
      for (auto& o : out) o = GetRandomActorOutput();
      return out;
    }
    std::vector<float> s(states_batch.size() * state_size_);
    for (size_t n = 0; n < states_batch.size(); ++n)
      std::copy(states_batch[n][0]->begin(), states_batch[n][0]->end(), s.begin() + n * state_size_);
    DQN_CK(dqnhip_select_actions(h_, s.data(), (int)states_batch.size(), out[0].data()));
    return out;
  }

  float EvaluateAction(const InputStates& input_states, const ActorOutput& action) {   // src/dqn.cpp:688-693
    float q = 0;
    DQN_CK(dqnhip_critic_forward(h_, DQNHIP_CRITIC, input_states[0]->data(), action.data(), 1, &q));
    return q;
  }

  void AddTransition(const Transition& t) {                                   // src/dqn.cpp:768-773
    const auto& nx = std::get<4>(t);
    DQN_CK(dqnhip_add_transition(h_, std::get<0>(t)[0]->data(), std::get<1>(t).data(), std::get<2>(t), std::get<3>(t),
                                 nx ? (*nx)->data() : nullptr, nx ? 0 : 1));
  }

  void AddTransitions(const std::vector<Transition>& ts) {                    // src/dqn.cpp:775-781
    const size_t n = ts.size(), S = state_size_;
    std::vector<float> s(n * S), nx(n * S, 0.f), a(n * 10), r(n), mc(n);
    std::vector<uint8_t> term(n);
    for (size_t i = 0; i < n; ++i) {
      std::copy(std::get<0>(ts[i])[0]->begin(), std::get<0>(ts[i])[0]->end(), s.begin() + i * S);
      std::copy(std::get<1>(ts[i]).begin(), std::get<1>(ts[i]).end(), a.begin() + i * 10);
      r[i] = std::get<2>(ts[i]); mc[i] = std::get<3>(ts[i]);
      const auto& n4 = std::get<4>(ts[i]);
      term[i] = n4 ? 0 : 1;                                                   // terminal <=> next_state == none (:878)
      if (n4) std::copy((*n4)->begin(), (*n4)->end(), nx.begin() + i * S);
    }
    DQN_CK(dqnhip_add_transitions(h_, s.data(), a.data(), r.data(), mc.data(), nx.data(), term.data(), (int)n));
  }

  void LabelTransitions(std::vector<Transition>& ts) {                        // src/dqn.cpp:783-797
    if (ts.empty()) DQN_FATAL("Need at least one transition to label.");
    std::vector<float> r(ts.size()), mc(ts.size());
    for (size_t i = 0; i < ts.size(); ++i) r[i] = std::get<2>(ts[i]);
    DQN_CK(dqnhip_label_transitions(gamma_, r.data(), (int)ts.size(), mc.data()));
    for (size_t i = 0; i < ts.size(); ++i) std::get<3>(ts[i]) = mc[i];
  }

  void Update() {                                                             // src/dqn.cpp:799-826
    if (memory_size() < flags_.memory_threshold) return;
    std::pair<float, float> res = UpdateActorCritic();
    if (critic_iter() % flags_.loss_display_iter == 0) {
      std::printf("[Agent%d] Critic Iteration %d, loss = %g\n", tid_, critic_iter(), smoothed_critic_loss_);
      smoothed_critic_loss_ = 0;
    }
    smoothed_critic_loss_ += res.first / float(flags_.loss_display_iter);
    if (actor_iter() % flags_.loss_display_iter == 0) {
      std::printf("[Agent%d] Actor Iteration %d, avg_q_value = %g\n", tid_, actor_iter(), smoothed_actor_loss_);
      smoothed_actor_loss_ = 0;
    }
    smoothed_actor_loss_ += res.second / float(flags_.loss_display_iter);
    const bool critic_needs_snapshot = critic_iter() >= last_snapshot_iter_ + flags_.snapshot_freq;   // :818-825
    const bool actor_needs_snapshot = actor_iter() >= last_snapshot_iter_ + flags_.snapshot_freq;
    if (critic_needs_snapshot || actor_needs_snapshot) { Snapshot(); last_snapshot_iter_ = max_iter(); }
  }

  // Loading methods (src/dqn.hpp:66-71, src/dqn.cpp:525-557, 1180-1226)
  void RestoreActorSolver(const std::string& f) { DQN_CK(dqnhip_solver_restore(h_, DQNHIP_ACTOR, f.c_str())); last_snapshot_iter_ = max_iter(); }
  void RestoreCriticSolver(const std::string& f) { DQN_CK(dqnhip_solver_restore(h_, DQNHIP_CRITIC, f.c_str())); last_snapshot_iter_ = max_iter(); }
  void LoadActorWeights(const std::string& f) { DQN_CK(dqnhip_load_caffemodel(h_, DQNHIP_ACTOR, f.c_str())); }
  void LoadCriticWeights(const std::string& f) { DQN_CK(dqnhip_load_caffemodel(h_, DQNHIP_CRITIC, f.c_str())); }
  void LoadReplayMemory(const std::string& f) { DQN_CK(dqnhip_load_replay_memory(h_, f.c_str())); }
  void SnapshotReplayMemory(const std::string& f) { DQN_CK(dqnhip_snapshot_replay_memory(h_, f.c_str())); }

  // Share the parameters of the first layers / the replay memory with a teammate
  // (src/dqn.hpp:121-124, src/dqn.cpp:1047-1083); this learner stays the owner
  void ShareParameters(DQN& other, int num_actor_layers_to_share, int num_critic_layers_to_share) {
    DQN_CK(dqnhip_share_parameters(h_, other.h_, num_actor_layers_to_share, num_critic_layers_to_share));
  }
  void ShareReplayMemory(DQN& other) { DQN_CK(dqnhip_share_replay_memory(h_, other.h_)); }

  // Snapshot the model/solver/replay memory (src/dqn.cpp:582-620)
  void Snapshot() { Snapshot(save_path_, flags_.remove_old_snapshots, flags_.snapshot_memory); }
  void Snapshot(const std::string& snapshot_prefix, bool remove_old = false, bool snapshot_memory = true) {
    DQN_CK(dqnhip_snapshot(h_, save_path_.c_str(), snapshot_prefix.c_str(), remove_old, snapshot_memory));
  }

  // Converts an ActorOutput into an action by sampling over discrete actions (src/dqn.cpp:180-194)
  Action SampleAction(const ActorOutput& actor_output) {
    float dash_prob = std::max(0., actor_output[DASH] + 1.0);
    float turn_prob = std::max(0., actor_output[TURN] + 1.0);
    float tackle_prob = 0;                                  // Remove tackle action
    float kick_prob = std::max(0., actor_output[KICK] + 1.0);
    std::discrete_distribution<int> dist{dash_prob, turn_prob, tackle_prob, kick_prob};
    action_t max_act = (action_t)dist(random_engine);
    Action action;
    action.action = max_act;
    action.arg1 = actor_output[kActionSize + GetParamOffset(max_act, 0)];
    const int arg2_offset = GetParamOffset(max_act, 1);
    action.arg2 = arg2_offset < 0 ? 0 : actor_output[kActionSize + arg2_offset];
    return action;
  }

  void ClearReplayMemory() { DQN_CK(dqnhip_clear_memory(h_)); }
  int memory_size() const { int32_t n = 0; DQN_CK(dqnhip_memory_size(h_, &n)); return n; }
  int min_iter() const { return std::min(actor_iter(), critic_iter()); }
  int max_iter() const { return std::max(actor_iter(), critic_iter()); }
  int critic_iter() const { int32_t a, c; DQN_CK(dqnhip_get_iters(h_, &a, &c)); return c; }
  int actor_iter() const { int32_t a, c; DQN_CK(dqnhip_get_iters(h_, &a, &c)); return a; }
  int state_size() const { return state_size_; }
  const std::string& save_path() const { return save_path_; }
  int unum() const { return unum_; }
  void set_unum(int unum) { unum_ = unum; }
  dqnhip_handle handle() const { return h_; }

  // protected in the reference; public here so tests can drive it with explicit indices
  std::pair<float, float> UpdateActorCritic() {                               // src/dqn.cpp:828-972
    // SampleTransitionsFromMemory (:501-509) on the host std::mt19937, as the reference does
    std::vector<int32_t> idx(flags_.minibatch);
    const int size = memory_size();
    for (auto& i : idx) i = std::uniform_int_distribution<int>(0, size - 1)(random_engine);
    float loss = 0, avgq = 0;
    DQN_CK(dqnhip_update(h_, idx.data(), &loss, &avgq));
    return std::make_pair(loss, avgq);
  }

 protected:
  Flags flags_;
  const double gamma_;
  std::mt19937 random_engine;
  float smoothed_critic_loss_ = 0, smoothed_actor_loss_ = 0;
  int last_snapshot_iter_ = 0;
  std::string save_path_;
  const int state_size_;
  int tid_;
  int unum_ = 0;
  dqnhip_handle h_ = nullptr;
};

}  // namespace dqn

#endif  // DQN_ADAPTOR_HPP_
