// Drives include/dqn.hpp's `dqn::DQN` (+ dqn-hfo_amd/csrc/dqn_dropin.cpp) the way src/dqn_main.cpp drives
// the reference class (solver parameters :221-262, PlayOneEpisode :97-153, the update burst :357-363,
// resume :213-220 + 268-286, sharing :305-323), with a synthetic environment and a narrower tower taken
// from an edited prototxt.  Built by tests/test_cpp_adaptor.py with g++ against libdqnhip.so and
// include/shim/ (no reference sources involved); run only on the GPU box.
#include <cmath>
#include <cstdio>

#include <gflags/gflags.h>

#include "dqn.hpp"

using namespace hfo;

// a subclass reaches the protected update, as a white-box test of the reference would
struct TestDQN : dqn::DQN {
  using dqn::DQN::DQN;
  std::pair<float, float> Step() { return UpdateActorCritic(); }
  std::vector<dqn::InputStates> States(int n) { return SampleStatesFromMemory(n); }
};

// -check cpu_mode: construct under Caffe CPU mode (the driver's -gpu=false, src/dqn_main.cpp:208-212) -> the adaptor must stop with its
//                  message before touching the device;  -check select_cap: SelectActions on minibatch + 1 states -> the reference's
//                  CHECK_LE (src/dqn.cpp:699) unless -select_actions_cap widens it
DEFINE_string(check, "", "run one boundary-behaviour check instead of the episode loop: cpu_mode | select_cap");

int main(int argc, char** argv) {
  gflags::ParseCommandLineFlags(&argc, &argv, true);     // the learner flags are defined by dqn_dropin.cpp
  caffe::Caffe::set_mode(FLAGS_check == "cpu_mode" ? caffe::Caffe::CPU : caffe::Caffe::GPU);
  const int num_features = 59;                      // NumStateFeatures(1), src/hfo_game.hpp:13-16
  caffe::SolverParameter actor_sp, critic_sp;
  // tower 128-64-64-64: what a user gets by editing <save>_actor.prototxt / _critic.prototxt
  const int widths[4] = {128, 64, 64, 64};
  caffe::NetParameter an = dqn::CreateActorNet(num_features), cn = dqn::CreateCriticNet(num_features);
  for (caffe::NetParameter* np : {&an, &cn})
    for (int i = 0, k = 0; i < np->layer_size(); ++i)
      if (np->layer(i).type() == "InnerProduct" && np->layer(i).name().rfind("ip", 0) == 0)
        np->mutable_layer(i)->mutable_inner_product_param()->set_num_output(widths[k++]);
  actor_sp.mutable_net_param()->CopyFrom(an); critic_sp.mutable_net_param()->CopyFrom(cn);
  for (caffe::SolverParameter* sp : {&actor_sp, &critic_sp}) {
    sp->set_type("Adam"); sp->set_momentum(.95f); sp->set_momentum2(.999f); sp->set_clip_gradients(10); sp->set_lr_policy("fixed");
  }
  actor_sp.set_base_lr(1e-5f); critic_sp.set_base_lr(1e-3f);
  TestDQN dqn(actor_sp, critic_sp, "/tmp/dqnhip_adaptor_smoke_run_agent0", num_features, 0);
  std::mt19937 env(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  if (FLAGS_check == "cpu_mode") { std::fprintf(stderr, "the constructor accepted Caffe CPU mode\n"); return 20; }
  if (FLAGS_check == "select_cap") {
    std::vector<dqn::InputStates> batch(dqn.minibatch_size() + 1);
    for (auto& in : batch) { in[0] = std::make_shared<dqn::StateData>(num_features); for (auto& v : *in[0]) v = U(env); }
    const auto out = dqn.SelectActions(batch, 0.0);       // aborts here under the default cap
    std::printf("select_cap: %zu actions\n", out.size());
    return out.size() == batch.size() ? 0 : 21;
  }
  int total_steps = 0;
  for (int episode = 0; episode < 6; ++episode) {
    std::vector<dqn::Transition> ep;
    auto state = std::make_shared<dqn::StateData>(num_features);
    for (auto& v : *state) v = U(env);
    const int len = 40 + 5 * episode;
    for (int t = 0; t < len; ++t) {
      dqn::InputStates in = {{state}};
      dqn::ActorOutput ao = dqn.SelectAction(in, 0.5);
      Action act = dqn::GetAction(ao);
      if (act.action == TACKLE) { std::fprintf(stderr, "GetAction returned TACKLE\n"); return 1; }
      auto next = std::make_shared<dqn::StateData>(num_features);
      for (auto& v : *next) v = U(env);
      const float reward = 0.1f * U(env);
      if (t + 1 < len) ep.emplace_back(in, ao, reward, 0.f, next);
      else ep.emplace_back(in, ao, reward + 5.f, 0.f, boost::none);
      state = next;
    }
    dqn.LabelTransitions(ep);
    dqn.AddTransitions(ep);
    total_steps += len;
    const int n_updates = int(len * 0.1);           // FLAGS_update_ratio, src/dqn_main.cpp:358
    for (int i = 0; i < n_updates; ++i) dqn.Update();
  }
  if (dqn.memory_size() != total_steps) { std::fprintf(stderr, "memory_size %d != %d\n", dqn.memory_size(), total_steps); return 2; }
  if (dqn.actor_iter() < 10 || dqn.actor_iter() != dqn.critic_iter()) { std::fprintf(stderr, "iters %d %d\n", dqn.actor_iter(), dqn.critic_iter()); return 3; }
  // snapshot + resume through the adaptor, the way dqn_main.cpp does at start-up (:213-220, 268-286)
  const std::string prefix = "/tmp/dqnhip_adaptor_smoke_agent0";
  dqn::RemoveFilesMatchingRegexp(prefix + "_.*");
  dqn.Snapshot(prefix, false, true);
  std::string a_snap, c_snap, m_snap;
  dqn::FindLatestSnapshot(prefix, a_snap, c_snap, m_snap);
  if (a_snap.empty() || c_snap.empty() || m_snap.empty()) { std::fprintf(stderr, "FindLatestSnapshot found nothing\n"); return 5; }
  {
    dqn::DQN resumed(actor_sp, critic_sp, prefix, num_features, 0);
    resumed.RestoreActorSolver(a_snap); resumed.RestoreCriticSolver(c_snap); resumed.LoadReplayMemory(m_snap);
    if (resumed.actor_iter() != dqn.actor_iter() || resumed.memory_size() != dqn.memory_size()) return 6;
    resumed.Update();
  }
  {
    // a teammate sharing the first layers and the replay memory (src/dqn_main.cpp:305-323)
    dqn::DQN mate(actor_sp, critic_sp, prefix + "_mate", num_features, 1);
    dqn.ShareParameters(mate, 2, 1);
    dqn.ShareReplayMemory(mate);
    if (mate.memory_size() != dqn.memory_size()) return 8;
    const int it = dqn.actor_iter();
    mate.Update();                                   // the teammate's solver writes the shared layers
    if (dqn.actor_iter() != it || mate.actor_iter() != 1) return 9;
    dqn.Update();
  }
  dqn::RemoveFilesMatchingRegexp(prefix + "_.*");
  Action sampled = dqn.SampleAction(dqn.GetRandomActorOutput());
  if (sampled.action == TACKLE) return 7;
  auto res = dqn.Step();
  if (!std::isfinite(res.first) || !std::isfinite(res.second)) return 4;
  if (dqn.States(5).size() != 5 || (int)dqn.States(3)[2][0]->size() != num_features) return 10;
  if (dqn::FilesMatchingRegexp("/tmp/dqnhip_adaptor_smoke_agent0_.*").size() != 0) return 11;
  std::printf("adaptor smoke OK: %d transitions, %d updates, loss %.9g avg_q %.9g\n", dqn.memory_size(), dqn.actor_iter(), res.first, res.second);
  return 0;
}
