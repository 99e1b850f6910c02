/*
 * dqnhip_env.h — batched environment front-end for the learner of dqnhip.h.
 *
 * Replaces, for N concurrent workers whose state vectors live in HBM, the per-agent episode
 * loop of the reference's src/dqn_main.cpp:97-153 (PlayOneEpisode) on the learner side:
 *     SelectAction(state, epsilon)            src/dqn.cpp:684-711  (one epsilon draw per call)
 *     GetAction(actor_output)                 src/dqn.cpp:196-208
 *     HFOGameState::update + reward           src/hfo_game.cpp:122-236
 *     Transition(state, actor_output, reward, 0, next | none)   src/dqn_main.cpp:138-141
 *     LabelTransitions + AddTransitions at episode end          src/dqn_main.cpp:145-150
 * The HFO server itself (rcssserver over UDP inside hfo.step()) is outside the reference repo
 * and absent here; the state stream it would deliver is replaced by a SYNTHETIC generator
 * (features U(-1,1), indices 12/54 in {-1,+1}, (13,14)/(51,52) = (sin,cos) of a
 * uniform angle, geometric episode length capped at --frames-per-trial) driven by the same
 * counter-based RNG on the device and in the CPU oracle, so the two can be compared
 * transition by transition.  Everything downstream of the state stream is the reference's
 * learner-side logic.
 */
#ifndef DQNHIP_ENV_H_
#define DQNHIP_ENV_H_

#include <stdint.h>

#include "dqnhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dqnhip_env_config {
  int32_t struct_size;   /* = sizeof(dqnhip_env_config) */
  int32_t workers;       /* concurrent HFO workers (agents) feeding this learner's replay */
  int32_t max_steps;     /* --frames-per-trial (500, src/hfo_game.cpp:8): forced OUT_OF_TIME */
  int32_t unum;          /* our uniform number (HFOGameState::our_unum) */
  float p_end;           /* per-step probability that the synthetic episode ends (1/mean length) */
  float p_goal;          /* given an end, probability that the status is GOAL */
  uint64_t seed;         /* key of the counter-based generator */
} dqnhip_env_config;

typedef struct dqnhip_env* dqnhip_env_handle;

/* Creates N workers attached to learner `h` (their transitions go to its replay ring) and
 * resets every worker: first synthetic state, HFOGameState() + the initial update that
 * PlayOneEpisode performs after its forced DASH(0,0) (src/dqn_main.cpp:103-105). */
int dqnhip_env_create(dqnhip_handle h, const dqnhip_env_config* cfg, dqnhip_env_handle* out);
int dqnhip_env_destroy(dqnhip_env_handle e);

/* Advance every worker by n_steps environment steps (N * n_steps transitions).  Per step:
 * one batched actor forward over the N current states, per-worker epsilon draw (random
 * ActorOutput with GetRandomActorOutput's ranges, src/dqn.cpp:664-682, or the greedy
 * output), GetAction, synthetic next state + status, HFOGameState update + reward,
 * transition appended to the worker's episode; finished episodes are labelled
 * (LabelTransitions) and appended to the replay ring in worker order (AddTransitions).
 * Asynchronous on the learner's stream. */
int dqnhip_env_step(dqnhip_env_handle e, float epsilon, int32_t n_steps);

/* Blocks; totals since creation: transitions generated, episodes finished, sum of rewards,
 * goals.  Also refreshes the learner's host-side memory_size. */
int dqnhip_env_stats(dqnhip_env_handle e, int64_t* env_steps, int64_t* episodes, double* reward_sum,
                     int64_t* goals);

/* Parity / debugging: per-worker values of the LAST step.  name: "action" [N] (GetAction index),
 * "arg1" [N], "arg2" [N], "reward" [N], "actor_out" [N,10], "state" [N,S] (current state),
 * "episode_len" [N]. */
int dqnhip_env_debug_read(dqnhip_env_handle e, const char* name, float* host, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* DQNHIP_ENV_H_ */
