// include/shim/HFO.hpp — a SYNTHETIC stand-in for LARG/HFO's agent-side header, for boxes
// without HFO / rcssserver (this image).  It provides the names the reference
// uses (src/dqn_main.cpp:97-153, 288-290; src/hfo_game.cpp:69-78, 122-236; src/dqn.cpp:181-186,
// 210-215): hfo::action_t (DASH..KICK = 0..3 are pinned by the reference itself, the rest follow
// upstream HFO's order), status_t, Player, LEFT, LOW_LEVEL_FEATURE_SET, ActionToString and an
// HFOEnvironment whose "server" is a per-agent random process:
//   * state vectors as SURVEY.md §8d prescribes: i.i.d. U(-1,1); features 12 and 54 in {-1,+1};
//     (13,14) and (51,52) = (sin, cos) of an angle U(-pi,pi) so acos() is well defined
//     (src/hfo_game.cpp:138-143);
//   * an episode ends at each step with probability HFO_SHIM_P_END (default 0.02) or after
//     HFO_SHIM_FRAMES steps (default 500 = --frames-per-trial, src/hfo_game.cpp:8); it ends as a
//     GOAL with probability HFO_SHIM_P_GOAL (default 0.3), else OUT_OF_BOUNDS / CAPTURED_BY_DEFENSE
//     / OUT_OF_TIME;
//   * the number of features is HFO_SHIM_FEATURES (default 59 = NumStateFeatures(1)).
// Nothing here is on the learner's path; it only lets the unchanged driver run end to end.
#ifndef DQNHIP_SHIM_HFO_HPP_
#define DQNHIP_SHIM_HFO_HPP_

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

namespace hfo {

enum action_t {
  DASH = 0, TURN = 1, TACKLE = 2, KICK = 3,      // pinned by src/dqn.cpp:181-186, 210-215
  KICK_TO, MOVE_TO, DRIBBLE_TO, INTERCEPT, MOVE, SHOOT, PASS, DRIBBLE, CATCH, NOOP, QUIT,
  REDUCE_ANGLE_TO_GOAL, MARK_PLAYER, DEFEND_GOAL, GO_TO_BALL
};
enum status_t { IN_GAME = 0, GOAL, CAPTURED_BY_DEFENSE, OUT_OF_BOUNDS, OUT_OF_TIME, SERVER_DOWN };
enum side_t { RIGHT = -1, NEUTRAL = 0, LEFT = 1 };
enum feature_set_t { LOW_LEVEL_FEATURE_SET, HIGH_LEVEL_FEATURE_SET };
struct Player { side_t side; int unum; };

inline std::string ActionToString(action_t a) {
  static const char* n[] = {"Dash", "Turn", "Tackle", "Kick", "KickTo", "MoveTo", "DribbleTo", "Intercept", "Move", "Shoot",
                            "Pass", "Dribble", "Catch", "No-op", "Quit", "Reduce_Angle_To_Goal", "Mark_Player", "Defend_Goal", "Go_To_Ball"};
  return (int)a >= 0 && (int)a < (int)(sizeof n / sizeof n[0]) ? n[a] : "Unknown";
}

namespace shim {
inline double env_num(const char* name, double dflt) { const char* e = std::getenv(name); return e ? std::atof(e) : dflt; }
inline std::atomic<int>& next_unum() { static std::atomic<int> u{7}; return u; }      // first agent: uniform number 7, then 8, ...
}  // namespace shim

class HFOEnvironment {
 public:
  HFOEnvironment()
      : n_features_((int)shim::env_num("HFO_SHIM_FEATURES", 59)), frames_((int)shim::env_num("HFO_SHIM_FRAMES", 500)),
        p_end_(shim::env_num("HFO_SHIM_P_END", 0.02)), p_goal_(shim::env_num("HFO_SHIM_P_GOAL", 0.3)) {}

  void connectToServer(feature_set_t = LOW_LEVEL_FEATURE_SET, std::string /*config_dir*/ = "bin/teams/base/config/formations-dt",
                       int server_port = 6000, std::string /*server_addr*/ = "localhost", std::string /*team_name*/ = "base_left",
                       bool /*play_goalie*/ = false, std::string /*record_dir*/ = "") {
    unum_ = shim::next_unum()++;
    rng_.seed((unsigned)(server_port * 31 + unum_));
    state_.assign(n_features_, 0.0f);
    draw_state();
  }

  const std::vector<float>& getState() { return state_; }

  // upstream: void act(action_t, ...) (C varargs); call sites pass ints or floats (src/dqn_main.cpp:104, 131)
  template <class... Args>
  void act(action_t action, Args... /*params*/) { pending_ = action; }

  status_t step() {
    if (pending_ == QUIT) return status_;                                   // src/dqn_main.cpp:328-329: act(QUIT); step(); result unused
    if (status_ != IN_GAME) { status_ = IN_GAME; t_ = 0; }                 // previous step closed an episode: a new one starts
    ++t_;
    draw_state();
    std::uniform_real_distribution<double> U(0.0, 1.0);
    if (t_ >= frames_) status_ = OUT_OF_TIME;
    else if (t_ >= 2 && U(rng_) < p_end_) {             // never on the first step: src/dqn_main.cpp:106 CHECKs that
      const double u = U(rng_);
      status_ = u < p_goal_ ? GOAL : (u < p_goal_ + (1 - p_goal_) / 2 ? OUT_OF_BOUNDS : CAPTURED_BY_DEFENSE);
    }
    return status_;
  }

  int getUnum() { return unum_; }
  int getNumTeammates() { return 0; }
  int getNumOpponents() { return 0; }
  // the agent itself holds the ball (HFOGameState::EOT_reward CHECKs side == LEFT on a goal, src/hfo_game.cpp:208)
  Player playerOnBall() { return Player{LEFT, unum_}; }

 private:
  void draw_state() {
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (auto& v : state_) v = U(rng_);
    if (n_features_ >= 56) {
      state_[12] = U(rng_) < 0 ? -1.0f : 1.0f;
      state_[54] = U(rng_) < 0 ? -1.0f : 1.0f;
      const float a = 3.14159265f * U(rng_), b = 3.14159265f * U(rng_);
      state_[13] = std::sin(a); state_[14] = std::cos(a);
      state_[51] = std::sin(b); state_[52] = std::cos(b);
    }
  }
  int n_features_, frames_;
  double p_end_, p_goal_;
  int unum_ = 7, t_ = 0;
  action_t pending_ = NOOP;
  status_t status_ = IN_GAME;
  std::mt19937 rng_;
  std::vector<float> state_;
};

}  // namespace hfo

#endif  // DQNHIP_SHIM_HFO_HPP_
