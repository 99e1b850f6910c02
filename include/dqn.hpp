// include/dqn.hpp — the reference's `dqn::DQN` surface (src/dqn.hpp:16-244) over the MI355X-native
// learner: what src/dqn_main.cpp includes instead of the reference's src/dqn.hpp when src/dqn.cpp
// is replaced by dqn-hfo_amd/csrc/dqn_dropin.cpp + libdqnhip.so.  The driver
// compiles UNCHANGED against it (tests/test_dropin_driver.py builds and runs it).
//
// Same constants, type aliases, name constants, public methods and free functions, same ownership
// (the driver new/deletes the DQN; Transitions are taken by const-ref and copied — here: into the
// device-resident ring), same error convention (glog CHECK / LOG(FATAL) -> abort).  What differs is
// behind the surface: no caffe::Net / caffe::Solver members — one dqnhip_handle (include/dqnhip.h).
// The reference's protected helpers that took caffe::Net& arguments (SelectActionGreedily,
// CriticForward, InputDataIntoLayers, CloneNet, SoftUpdateNet: src/dqn.hpp:149-184) have no caller
// outside src/dqn.cpp and are not declared.  The PUBLIC ShareLayer(caffe::Layer&, caffe::Layer&)
// (src/dqn.hpp:115-116) is declared for source compatibility; no caffe::Layer object exists behind
// this surface (its only caller is ShareParameters, src/dqn.cpp:1064-1075), so calling it is a
// LOG(FATAL) that points at ShareParameters.
#ifndef DQNHIP_DQN_HPP_
#define DQNHIP_DQN_HPP_

#include <algorithm>
#include <array>
#include <memory>
#include <random>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include <HFO.hpp>
#include <caffe/caffe.hpp>
#include <boost/optional.hpp>

#if __has_include("hfo_game.hpp")
#include "hfo_game.hpp"            // the reference's own header (Action, HFOGameState, NumStateFeatures)
#else
struct Action { hfo::action_t action; float arg1; float arg2; };     // src/hfo_game.hpp:7-11
#endif

#include "dqnhip.h"

namespace dqn {

constexpr auto kStateInputCount = 1;        // src/dqn.hpp:18
constexpr auto kMinibatchSize = 32;         // :19  (the learner's actual minibatch is -minibatch, default 32)
constexpr auto kActionSize = 4;             // :20
constexpr auto kActionParamSize = 6;        // :21
constexpr auto kActionInputDataSize = kMinibatchSize * kActionSize;
constexpr auto kActionParamsInputDataSize = kMinibatchSize * kActionParamSize;
constexpr auto kTargetInputDataSize = kMinibatchSize * kActionSize;
constexpr auto kFilterInputDataSize = kMinibatchSize * kActionSize;

using ActorOutput = std::array<float, kActionSize + kActionParamSize>;
using StateData   = std::vector<float>;
using StateDataSp = std::shared_ptr<StateData>;
using InputStates = std::array<StateDataSp, kStateInputCount>;
using Transition  = std::tuple<InputStates, ActorOutput, float, float, boost::optional<StateDataSp>>;

// layer / blob names: they are the keys of the .caffemodel / .prototxt files (src/dqn.hpp:38-51)
constexpr auto state_input_layer_name         = "state_input_layer";
constexpr auto action_input_layer_name        = "action_input_layer";
constexpr auto action_params_input_layer_name = "action_params_input_layer";
constexpr auto target_input_layer_name        = "target_input_layer";
constexpr auto filter_input_layer_name        = "filter_input_layer";
constexpr auto q_values_layer_name            = "q_values_layer";
constexpr auto states_blob_name        = "states";
constexpr auto actions_blob_name       = "actions";
constexpr auto action_params_blob_name = "action_params";
constexpr auto targets_blob_name       = "target";
constexpr auto filter_blob_name        = "filter";
constexpr auto q_values_blob_name      = "q_values";
constexpr auto loss_blob_name          = "loss";

class DQN {
 public:
  // src/dqn.hpp:58-60.  Reads base_lr / momentum / momentum2 / delta / clip_gradients from the two
  // solver parameters and the tower widths from their net_param (the ip<i>_layer num_outputs of
  // CreateActorNet / CreateCriticNet or of a user-edited <save>_actor.prototxt).
  DQN(caffe::SolverParameter& actor_solver_param, caffe::SolverParameter& critic_solver_param,
      std::string save_path, int state_size, int tid);
  ~DQN();
  DQN(const DQN&) = delete;
  DQN& operator=(const DQN&) = delete;

  void Benchmark(int iterations = 1000);

  void RestoreActorSolver(const std::string& actor_solver);
  void RestoreCriticSolver(const std::string& critic_solver);
  void LoadActorWeights(const std::string& actor_model_file);
  void LoadCriticWeights(const std::string& critic_weights);
  void LoadReplayMemory(const std::string& filename);

  void Snapshot();
  void Snapshot(const std::string& snapshot_prefix, bool remove_old = false, bool snapshot_memory = true);

  ActorOutput GetRandomActorOutput();
  ActorOutput SelectAction(const InputStates& input_states, double epsilon);
  std::vector<ActorOutput> SelectActions(const std::vector<InputStates>& states_batch, double epsilon);
  Action SampleAction(const ActorOutput& actor_output);
  float EvaluateAction(const InputStates& input_states, const ActorOutput& action);

  void AddTransition(const Transition& transition);
  void AddTransitions(const std::vector<Transition>& transitions);
  void LabelTransitions(std::vector<Transition>& transitions);

  void Update();

  void ClearReplayMemory();
  void SnapshotReplayMemory(const std::string& filename);
  int memory_size() const;

  void ShareLayer(caffe::Layer<float>& param_owner, caffe::Layer<float>& param_slave);     // src/dqn.hpp:115-116
  void ShareParameters(DQN& other, int num_actor_layers_to_share, int num_critic_layers_to_share);
  void ShareReplayMemory(DQN& other);

  int min_iter() const { return std::min(actor_iter(), critic_iter()); }
  int max_iter() const { return std::max(actor_iter(), critic_iter()); }
  int critic_iter() const;
  int actor_iter() const;
  int state_size() const { return state_size_; }
  const std::string& save_path() const { return save_path_; }
  int unum() const { return unum_; }
  void set_unum(int unum) { unum_ = unum; }

  // ---- beyond the reference's surface ----
  dqnhip_handle handle() const { return h_; }          // the C-ABI learner (include/dqnhip.h)
  int minibatch_size() const { return minibatch_; }
  // UpdateActorCritic on caller-supplied sampled indices (parity tests: SURVEY.md F5)
  std::pair<float, float> UpdateActorCritic(const std::vector<int>& sampled_transitions);

 protected:
  std::pair<float, float> UpdateActorCritic();                              // src/dqn.cpp:828-972
  std::vector<int> SampleTransitionsFromMemory(int n);                      // :501-509, host std::mt19937
  std::vector<InputStates> SampleStatesFromMemory(int n);                   // :511-523

  caffe::SolverParameter actor_solver_param_;
  caffe::SolverParameter critic_solver_param_;
  const int replay_memory_capacity_;
  const double gamma_;
  std::mt19937 random_engine;
  float smoothed_critic_loss_, smoothed_actor_loss_;
  int last_snapshot_iter_;
  std::string save_path_;
  const int state_size_;
  int tid_;
  int unum_;
  int minibatch_;
  bool dp_ = false;                                    // -dp_rendezvous given: UpdateActorCritic() is one rank's share of a data-parallel update
  bool dp_sync_pending_ = false;                       // a Restore* / Load* ran since the group's last broadcast: re-sync at the next update
  // UpdateActorCritic(): the next call's indices, predicted from a copy of random_engine (dqnhip_update_chained)
  bool spec_valid_ = false; int spec_size_ = 0;
  std::mt19937 spec_before_, spec_after_;              // the engine as the prediction found / left it
  std::vector<int> spec_idx_;
  bool dp_synced_once_ = false;                        // the group's first update has passed: Restore* / Load* are refused from here on (one-sided collective)
  void SyncReplicasIfPending();
  void RearmReplicaSync(const char* what);
  dqnhip_handle h_;
};

caffe::NetParameter CreateActorNet(int state_size);       // src/dqn.hpp:204, src/dqn.cpp:418-429
caffe::NetParameter CreateCriticNet(int state_size);      // src/dqn.hpp:205, src/dqn.cpp:431-454

Action GetAction(const ActorOutput& actor_output);
std::vector<std::string> FilesMatchingRegexp(const std::string& regexp);
void RemoveFilesMatchingRegexp(const std::string& regexp);
void RemoveSnapshots(const std::string& regexp, int min_iter);
void FindLatestSnapshot(const std::string& snapshot_prefix, std::string& actor_snapshot,
                        std::string& critic_snapshot, std::string& memory_snapshot);
int FindHiScore(const std::string& snapshot_prefix);
std::string PrintActorOutput(const ActorOutput& actor_output);
int GetParamOffset(const hfo::action_t action, const int arg_num = 0);     // src/dqn.cpp:162-178 (file-local there)

}  // namespace dqn

#endif  // DQNHIP_DQN_HPP_
