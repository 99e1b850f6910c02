// Drives the C++ `dqn::DQN` adaptor the way src/dqn_main.cpp drives the reference class
// (PlayOneEpisode :97-153 + the update burst :357-363), with a synthetic environment.
// Built by tests/test_cpp_adaptor.py with g++ against libdqnhip.so; run only on the GPU box.
#include <cmath>
#include <cstdio>

#include "../../dqn-hfo_amd/csrc/dqn_adaptor.hpp"

int main() {
  dqn::Flags flags;
  flags.seed = 7; flags.memory = 5000; flags.memory_threshold = 100; flags.hidden = {128, 64, 64, 64};
  dqn::SolverParams actor_sp, critic_sp;
  actor_sp.base_lr = 1e-5f;
  const int num_features = 59;                      // NumStateFeatures(1), src/hfo_game.hpp:13-16
  dqn::DQN dqn(actor_sp, critic_sp, "/tmp/dqnhip_adaptor_smoke_run_agent0", num_features, 0, flags);
  std::mt19937 env(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
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
      else ep.emplace_back(in, ao, reward + 5.f, 0.f, std::nullopt);
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
    dqn::DQN resumed(actor_sp, critic_sp, prefix, num_features, 0, flags);
    resumed.RestoreActorSolver(a_snap); resumed.RestoreCriticSolver(c_snap); resumed.LoadReplayMemory(m_snap);
    if (resumed.actor_iter() != dqn.actor_iter() || resumed.memory_size() != dqn.memory_size()) return 6;
    resumed.Update();
  }
  {
    // a teammate sharing the first layers and the replay memory (src/dqn_main.cpp:305-323)
    dqn::DQN mate(actor_sp, critic_sp, prefix + "_mate", num_features, 1, flags);
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
  auto res = dqn.UpdateActorCritic();
  if (!std::isfinite(res.first) || !std::isfinite(res.second)) return 4;
  std::printf("adaptor smoke OK: %d transitions, %d updates, loss %g avg_q %g\n", dqn.memory_size(), dqn.actor_iter(), res.first, res.second);
  return 0;
}
