// env.hip.h — device side of the batched env front-end (include/dqnhip_env.h): per-worker
// epsilon-greedy selection + GetAction, synthetic HFO state stream, HFOGameState reward
// shaping, per-episode buffers, LabelTransitions + AddTransitions at episode end.
// Reference lines: src/dqn_main.cpp:97-153, src/dqn.cpp:162-208, 664-711, 768-797,
// src/hfo_game.cpp:109-236.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "small_kernels.hip.h"

namespace dqnhip {

// HFOGameState (src/hfo_game.hpp:29-60), one per worker, SoA-free for clarity (N is small)
struct GameState {
  float old_ball_prox, ball_prox_delta, old_kickable, kickable_delta, old_ball_dist_goal, ball_dist_goal_delta;
  int steps, episode_over, got_kickable_reward, pass_active, pob, old_pob, status, pad;
};

struct EnvDev {
  int N, S, SP, T;                 // workers, state size, padded state size, max steps per episode
  int unum;
  float p_end, p_goal;
  unsigned long long seed;
  float* cur;                      // [Npad][SP] current states == actor input panel
  float* out16;                    // [Npad][16] greedy actor outputs of this step
  float* ep_s;                     // [N][T][SP]
  float* ep_a;                     // [N][T][16]
  float* ep_r;                     // [N][T]
  GameState* game;                 // [N]
  int* len;                        // [N] transitions in the open episode
  int* done;                       // [N] length of the episode that finished in this step, 0 if none
  unsigned long long* g;           // [N] per-worker draw counter
  // per-worker last-step values (debug / parity)
  int* act; float* arg1; float* arg2; float* rew;
  // totals
  unsigned long long* n_steps; unsigned long long* n_episodes; unsigned long long* n_goals; double* reward_sum;
  const float* eps;                // device scalar: this call's epsilon (not a kernel argument, so a captured step replays)
  // fused actor heads (action_layer + actionpara_layer on this worker's tower-top row): null -> out16 was
  // written by a separate head launch
  const float* head_x; const float* head_w; const float* head_b; int head_h;
  int* commit_ticket;              // arrival counter of k_env_flush (its last block publishes the ring bookkeeping); null: k_env_commit does
};

__device__ __forceinline__ float env_u01(unsigned long long seed, unsigned long long g, int w, int k) {
  return (float)(philox_u32(seed, g, (uint32_t)(w * 256 + k)) >> 8) * (1.0f / 16777216.0f);
}

// synthetic low-level feature f of worker w at draw counter g
__device__ __forceinline__ float env_feature(const EnvDev& e, unsigned long long g, int w, int f) {
  const float kPi = 3.14159265358979323846f;
  if (f == 13 || f == 14) { const float th = fmaf(2.0f, env_u01(e.seed, g, w, 16 + 13), -1.0f) * kPi; return f == 13 ? sinf(th) : cosf(th); }
  if (f == 51 || f == 52) { const float th = fmaf(2.0f, env_u01(e.seed, g, w, 16 + 51), -1.0f) * kPi; return f == 51 ? sinf(th) : cosf(th); }
  const float u = env_u01(e.seed, g, w, 16 + f);
  if (f == 12 || f == 54) return u < 0.5f ? -1.0f : 1.0f;
  return fmaf(2.0f, u, -1.0f);
}

// the angle of a (sin, cos) feature pair as HFOGameState::update forms it (src/hfo_game.cpp:137-144):
// acos of the cosine in double, negated when the sine is negative
__device__ __forceinline__ float game_angle(float sin_v, float cos_v) {
  float a = (float)acos((double)cos_v);
  if (sin_v < 0) a = (float)((double)a * -1.);
  return a;
}
// HFOGameState::update (src/hfo_game.cpp:122-173) minus hfo.step(): status / player_on_ball given.  The two angles
// (ball: features 51/52, goal: 13/14) come in precomputed: two software double acos are the longest dependent
// chain of a worker's step, and two lanes of the wave evaluate them side by side.
__device__ __forceinline__ void game_update(GameState& g, const float* st, int status, int pob, float ball_ang_rad, float goal_ang_rad) {
  g.status = status;
  if (status != 0) g.episode_over = 1;
  const float ball_proximity = st[53], goal_proximity = st[15];
  const float ball_dist = (float)(1.0 - (double)ball_proximity), goal_dist = (float)(1.0 - (double)goal_proximity);
  const float kickable = st[12];
  const float alpha = fmaxf(ball_ang_rad, goal_ang_rad) - fminf(ball_ang_rad, goal_ang_rad);
  const float ball_dist_goal = (float)sqrt((double)(ball_dist * ball_dist + goal_dist * goal_dist) -
                                           2. * (double)ball_dist * (double)goal_dist * cos((double)alpha));
  const float ball_vel_valid = st[54], ball_vel = st[55];
  if (ball_vel_valid != 0.0f && (double)ball_vel > -.5) g.pass_active = 1;   // kPassVelThreshold, src/hfo_game.hpp:18
  if (g.steps > 0) {
    g.ball_prox_delta = ball_proximity - g.old_ball_prox;
    g.kickable_delta = kickable - g.old_kickable;
    g.ball_dist_goal_delta = ball_dist_goal - g.old_ball_dist_goal;
  }
  g.old_ball_prox = ball_proximity; g.old_kickable = kickable; g.old_ball_dist_goal = ball_dist_goal;
  if (g.episode_over) { g.ball_prox_delta = 0; g.kickable_delta = 0; g.ball_dist_goal_delta = 0; }
  g.old_pob = g.pob; g.pob = pob;
  g.steps++;
}

// HFOGameState::reward (src/hfo_game.cpp:175-236): moveToBall + 3*kickToGoal + EOT; pass_reward()
// is evaluated for its side effect on pass_active but not added (:178-180)
__device__ __forceinline__ float game_reward(GameState& g, int our_unum, int* goal) {
  float mtb = 0;
  if (g.pob < 0 || g.pob == our_unum) mtb += g.ball_prox_delta;
  if (g.kickable_delta >= 1 && !g.got_kickable_reward) { mtb = (float)((double)mtb + 1.0); g.got_kickable_reward = 1; }
  float ktg = 0;
  if (g.pob == our_unum) ktg = -g.ball_dist_goal_delta;
  else if (g.got_kickable_reward) ktg = (float)(0.2 * (double)(-g.ball_dist_goal_delta));
  const float kickToGoal = (float)(3. * (double)ktg);
  if (g.pass_active && g.pob > 0 && g.pob != g.old_pob) g.pass_active = 0;
  float eot = 0;
  if (g.status == 1) { eot = (g.pob == our_unum) ? 5 : 1; *goal = 1; }
  return mtb + kickToGoal + eot;
}

__device__ __forceinline__ void game_reset(GameState& g) {
  g.old_ball_prox = 0; g.ball_prox_delta = 0; g.old_kickable = 0; g.kickable_delta = 0;
  g.old_ball_dist_goal = 0; g.ball_dist_goal_delta = 0; g.steps = 0; g.episode_over = 0;
  g.got_kickable_reward = 0; g.pass_active = 0; g.pob = 0; g.old_pob = 0; g.status = 0; g.pad = 0;
}

// new episode for worker w: first state, HFOGameState() and the initial update after the forced
// DASH(0,0) (src/dqn_main.cpp:103-105).  One wave; s_state is a [SP] LDS row.
__device__ __forceinline__ void env_reset_worker(const EnvDev& e, int w, int lane, float* s_state) {
  const unsigned long long g = e.g[w];
  for (int f = lane; f < e.SP; f += 64) {
    const float v = f < e.S ? env_feature(e, g, w, f) : 0.0f;
    s_state[f] = v; e.cur[(size_t)w * e.SP + f] = v;
  }
  __syncthreads();
  if (lane == 0) {
    GameState gs; game_reset(gs);
    game_update(gs, s_state, 0, 0, game_angle(s_state[51], s_state[52]), game_angle(s_state[13], s_state[14]));
    e.game[w] = gs; e.len[w] = 0; e.g[w] = g + 1;
  }
}

static __global__ void k_env_init(EnvDev e) {
  extern __shared__ float s_state[];
  env_reset_worker(e, blockIdx.x, threadIdx.x, s_state);
}

// one environment step of every worker: block = one worker, 4 waves.  What used to be one wave's serial chain
// (heads -> store -> reload -> features -> game update, 10 us at 64 workers) runs side by side: the four waves split
// the head dot products' k range, threads [0, SP) draw the next state while threads [SP, 2 SP) draw the first state of
// the next episode, and two lanes evaluate the two double-precision acos of HFOGameState::update: 8.4 us.
// (Also computing the first tower layer of the NEXT step here, from the state row the block has just produced — one
// launch fewer per step — was built and measured: 77 instead of 29 us per step at 64 workers.  64 blocks on 64 CUs
// each pull all of W0 (512 KB) through one CU's load path, where the separate launch spreads it over 256.)
static __global__ __launch_bounds__(256) void k_env_step(EnvDev e) {
  const float epsilon = e.eps[0];
  extern __shared__ float s_next[];                  // [2 SP]: next state | first state of the next episode
  __shared__ float s_ao[16];
  __shared__ float s_part[4][16];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = e.len[w];
  const unsigned long long g = e.g[w];
  GameState gs;                                      // fetched now, used at the end: its latency hides behind the heads
  if (tid == 0) gs = e.game[w];
  const float head_bias = (e.head_x != nullptr && tid < kNO) ? e.head_b[tid] : 0.0f;   // likewise
  // SelectAction(state, epsilon): ONE epsilon draw per call (src/dqn.cpp:700)
  const bool rnd = env_u01(e.seed, g, w, 0) < epsilon;
  if (e.head_x != nullptr) {
    // SelectActionGreedily's last step for this worker: the 10 head outputs of its tower-top row — the separate head
    // launch of the batched step folded in.  Thread t owns the float4 k-strip t (+ 256 strips per round).
    const float* x = e.head_x + (size_t)w * e.head_h;
    float acc[kNO];
#pragma unroll
    for (int j = 0; j < kNO; ++j) acc[j] = 0.0f;
    for (int k = tid * 4; k < e.head_h; k += 1024) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
#pragma unroll
      for (int j = 0; j < kNO; ++j) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(e.head_w + (size_t)j * e.head_h + k);
        acc[j] = fmaf(xv.x, wv.x, acc[j]); acc[j] = fmaf(xv.y, wv.y, acc[j]); acc[j] = fmaf(xv.z, wv.z, acc[j]); acc[j] = fmaf(xv.w, wv.w, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < kNO; ++j) { acc[j] = wave_sum64(acc[j]); if (lane == 0) s_part[wave][j] = acc[j]; }
  }
  float* eps_row = e.ep_s + ((size_t)w * e.T + len) * e.SP;
  float* cur = e.cur + (size_t)w * e.SP;
  // synthetic server: does the episode end on this step, and how (every thread evaluates the same two draws: the
  // block needs the answer to know which state the worker shows the actor next)
  int status = 0;
  if (env_u01(e.seed, g, w, 11) < e.p_end) status = env_u01(e.seed, g, w, 12) < e.p_goal ? 1 : 2;   // GOAL / CAPTURED_BY_DEFENSE
  if (status == 0 && len + 1 >= e.T) status = 4;                                                   // OUT_OF_TIME
  float* s_first = s_next + e.SP;                    // first state of the next episode (only if this one ends)
  for (int i = tid; i < 2 * e.SP; i += 256) {
    const bool first = i >= e.SP;
    const int f = first ? i - e.SP : i;
    if (!first) { eps_row[f] = cur[f]; s_next[f] = f < e.S ? env_feature(e, g, w, f) : 0.0f; }
    else if (status != 0) s_first[f] = f < e.S ? env_feature(e, g + 1, w, f) : 0.0f;
  }
  __syncthreads();
  if (tid < kAP) {
    float v = 0.0f;
    if (tid < kNO) {
      if (e.head_x != nullptr) v = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + head_bias;
      else v = e.out16[(size_t)w * kAP + tid];       // written by the separate head launch
    }
    if (e.head_x != nullptr) e.out16[(size_t)w * kAP + tid] = v;
    if (tid < kNO && rnd) {                          // GetRandomActorOutput (src/dqn.cpp:664-682)
      const float u = env_u01(e.seed, g, w, 1 + tid);
      if (tid < kNA) v = fmaf(2.0f, u, -1.0f);
      else if (tid == kNA + 0) v = fmaf(200.0f, u, -100.0f);
      else if (tid == kNA + 4) v = 100.0f * u;
      else v = fmaf(360.0f, u, -180.0f);
    }
    s_ao[tid] = v;
    e.ep_a[((size_t)w * e.T + len) * kAP + tid] = v;
  }
  // the actor's next input row: the next state, or — the episode is over — the first state of the new one
  // (src/dqn_main.cpp:97-105; the finished episode's transitions are labelled and added by k_env_flush, which
  // does not have to run before the next forward pass)
  const float* s_sel = status != 0 ? s_first : s_next;
  for (int f = tid; f < e.SP; f += 256) cur[f] = s_sel[f];     // (thread f also made the ep_s copy of cur[f] above)
  if (wave == 0) {
    // the four angles of this step's (at most) two HFOGameState::update calls, one per lane: lanes 0 / 1 the ball and
    // goal angles of the next state, lanes 2 / 3 those of the new episode's first state
    const float* src = (lane & 2) ? s_first : s_next;
    const bool ball = (lane & 1) == 0;
    float ang = 0.0f;
    if (lane < 2 || (lane < 4 && status != 0)) ang = game_angle(src[ball ? 51 : 13], src[ball ? 52 : 14]);
    const float goal_next = __shfl(ang, 1, 64), ball_first = __shfl(ang, 2, 64), goal_first = __shfl(ang, 3, 64);
    if (lane == 0) {
      // GetAction (src/dqn.cpp:196-208)
      float c0 = s_ao[0], c1 = s_ao[1], c3 = s_ao[3];
      const float c2 = -99999.0f;
      int best = 0; float bv = c0;
      if (c1 > bv) { best = 1; bv = c1; }
      if (c2 > bv) { best = 2; bv = c2; }
      if (c3 > bv) { best = 3; bv = c3; }
      const int o1 = best == 0 ? 0 : best == 1 ? 2 : best == 2 ? 3 : 4;
      const int o2 = best == 0 ? 1 : best == 3 ? 5 : -1;
      e.act[w] = best; e.arg1[w] = s_ao[kNA + o1]; e.arg2[w] = o2 < 0 ? 0.0f : s_ao[kNA + o2];
      const int pob = env_u01(e.seed, g, w, 13) < 0.5f ? e.unum : -1;
      game_update(gs, s_next, status, pob, ang, goal_next);
      int goal = 0;
      const float r = game_reward(gs, e.unum, &goal);
      e.ep_r[(size_t)w * e.T + len] = r; e.rew[w] = r;
      if (status != 0) {
        // new episode: HFOGameState() and the initial update after the forced DASH(0,0) (src/dqn_main.cpp:103-105)
        GameState g0; game_reset(g0);
        game_update(g0, s_first, 0, 0, ball_first, goal_first);
        e.game[w] = g0; e.len[w] = 0; e.g[w] = g + 2;
      } else {
        e.game[w] = gs; e.len[w] = len + 1; e.g[w] = g + 1;
      }
      e.done[w] = status != 0 ? len + 1 : 0;           // length of the finished episode (0: still open)
      e.n_steps[w] += 1; e.reward_sum[w] += (double)r; e.n_goals[w] += goal;
    }
  }
}

// the reference's deque arithmetic of AddTransitions(n) (src/dqn.cpp:775-781)
__device__ __forceinline__ void ring_add_plan(int cap, int n, int& head, int& size) {
  int pops = size + n - cap + 1;
  if (pops < 0) pops = 0;
  if (pops > size) pops = size;
  head = (int)(((long long)head + pops) % cap); size -= pops;
}

__device__ __forceinline__ void env_flush_worker(const EnvDev& e, const Ring& ring, const DevState* st, double gamma, int w, int len,
                                                 float* sm, int* s_start_p);
__device__ __forceinline__ void env_commit_body(const EnvDev& e, const Ring& ring, DevState* st);

// finished episodes: LabelTransitions + AddTransitions, in worker order (the worker itself was reset by k_env_step:
// this launch only reads done[] and the episode buffers, so it runs beside the next step's forward pass)
// one block of the flush: worker w of n_workers blocks (sm: [T] floats of LDS)
__device__ __forceinline__ void env_flush_block(const EnvDev& e, const Ring& ring, const DevState* st, double gamma, int w, int n_workers, float* sm) {
  __shared__ int s_start;
  const int len = e.done[w];
  if (len != 0) env_flush_worker(e, ring, st, gamma, w, len, sm, &s_start);
  if (e.commit_ticket == nullptr) return;
  // Every block has now read what it needs of done[] and (head,size); the LAST one to get here publishes the
  // ring bookkeeping (what k_env_commit did in its own launch).  It consumes nothing the other blocks of this
  // launch wrote, so the arrival count needs no fence.
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(e.commit_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == n_workers - 1);
    if (s_last) *e.commit_ticket = 0;
  }
  __syncthreads();
  if (s_last) env_commit_body(e, ring, const_cast<DevState*>(st));
}
static __global__ __launch_bounds__(256) void k_env_flush(EnvDev e, Ring ring, const DevState* st, double gamma) {
  extern __shared__ float sm[];          // [T] mc labels
  env_flush_block(e, ring, st, gamma, blockIdx.x, gridDim.x, sm);
}

__device__ __forceinline__ void env_flush_worker(const EnvDev& e, const Ring& ring, const DevState* st, double gamma, int w, int len,
                                                 float* sm, int* s_start_p) {
  int& s_start = *s_start_p;
  // AddTransitions of the workers before us, in worker order.  Every AddTransitions(n) (n <= cap-1)
  // moves the deque's tail by exactly n and leaves size = min(size + n, cap - 1), so this worker's
  // first slot is tail0 + (transitions flushed by lower-numbered workers): a parallel prefix sum
  // instead of replaying the deque arithmetic serially.
  __shared__ int s_pre[4];
  {
    int part = 0;
    for (int v = threadIdx.x; v < w; v += 256) part += e.done[v];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = part;
  }
  // rewards of the episode into LDS (all threads), so that the serial label scan below has no
  // global-memory latency inside its dependent chain
  for (int i = threadIdx.x; i < len; i += 256) sm[i] = e.ep_r[(size_t)w * e.T + i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    const long long tail0 = (long long)st->ring_head + st->ring_size;
    s_start = (int)((tail0 + ((s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]))) % ring.cap);
  }
  __syncthreads();
  const int start = s_start;
  const int SP = e.SP;                   // episode-buffer row = the actor's input panel width
  const int RSP = ring.SP;               // ring row (<= SP: the fp16 learner pads its panels to 128)
  if (wave == 0) {
    // LabelTransitions (src/dqn.cpp:783-797): reverse scan, gamma double, float store — the float rounding
    // of every step makes the chain inherently serial (one lane); it runs BESIDE the row copies of waves 1-3
    // instead of in front of them.  In place: sm[i] r -> mc
    if (lane == 0) {
      float target = sm[len - 1];
      int i = len - 2;
      for (; i >= 3; i -= 4) {
        const float r0 = sm[i], r1 = sm[i - 1], r2 = sm[i - 2], r3 = sm[i - 3];
        target = (float)((double)r0 + gamma * (double)target); sm[i] = target;
        target = (float)((double)r1 + gamma * (double)target); sm[i - 1] = target;
        target = (float)((double)r2 + gamma * (double)target); sm[i - 2] = target;
        target = (float)((double)r3 + gamma * (double)target); sm[i - 3] = target;
      }
      for (; i >= 0; --i) { target = (float)((double)sm[i] + gamma * (double)target); sm[i] = target; }
      e.n_episodes[w] += 1;
    }
  } else {
    // 192 threads copy the episode as float4s (RSP/4 = 16 or 32 per row: shifts, no division); the ring slot
    // needs one conditional subtraction (start < cap, t < len < cap) instead of a 64-bit modulo per element;
    // four independent float4 pairs in flight per thread
    const int q4 = RSP >> 2, sh = (q4 == 16) ? 4 : (q4 == 32 ? 5 : 0);
    const int tid3 = threadIdx.x - 64, n4 = len * q4;
    auto copy1 = [&](int idx) {
      const int t = sh ? (idx >> sh) : (idx / q4), c4 = sh ? (idx & (q4 - 1)) : (idx % q4);
      long long slot = (long long)start + t;
      if (slot >= ring.cap) slot -= ring.cap;
      const f32x4* src = reinterpret_cast<const f32x4*>(e.ep_s + ((size_t)w * e.T + t) * SP) + c4;
      const f32x4 sv = src[0];
      const f32x4 nv = (t + 1 < len) ? src[SP >> 2] : f32x4{0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(ring.state + slot * RSP)[c4] = sv;
      reinterpret_cast<f32x4*>(ring.next + slot * RSP)[c4] = nv;
    };
    int idx = tid3;
    for (; idx + 3 * 192 < n4; idx += 4 * 192) { copy1(idx); copy1(idx + 192); copy1(idx + 2 * 192); copy1(idx + 3 * 192); }
    for (; idx < n4; idx += 192) copy1(idx);
    for (int i = tid3; i < len * (kAP / 4); i += 192) {            // actor outputs: 4 float4 per transition
      const int t = i >> 2, c4 = i & 3;
      long long slot = (long long)start + t;
      if (slot >= ring.cap) slot -= ring.cap;
      reinterpret_cast<f32x4*>(ring.act + slot * kAP)[c4] = reinterpret_cast<const f32x4*>(e.ep_a + ((size_t)w * e.T + t) * kAP)[c4];
    }
    for (int t = tid3; t < len; t += 192) {
      long long slot = (long long)start + t;
      if (slot >= ring.cap) slot -= ring.cap;
      ring.reward[slot] = e.ep_r[(size_t)w * e.T + t];
      ring.term[slot] = (t + 1 < len) ? 0 : 1;                     // terminal <=> next_state == none
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < len; t += 256) {
    long long slot = (long long)start + t;
    if (slot >= ring.cap) slot -= ring.cap;
    ring.mc[slot] = sm[t];
  }
}

// publish the ring bookkeeping after all flushes of this step
template <int UNUSED = 0>
__global__ void k_set_float(float* p, float v) { *p = v; }

__device__ __forceinline__ void env_commit_body(const EnvDev& e, const Ring& ring, DevState* st) {
  // the closed form of the same sequence of AddTransitions (see k_env_flush): tail += total,
  // size = min(size + total, cap - 1)
  __shared__ int s_tot[4];
  int part = 0;
  for (int v = threadIdx.x; v < e.N; v += 256) { part += e.done[v]; e.done[v] = 0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const long long total = (long long)(s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
  if (total == 0) return;
  const long long tail = (long long)st->ring_head + st->ring_size + total;
  long long size = (long long)st->ring_size + total;
  if (size > ring.cap - 1) size = ring.cap - 1;
  st->ring_head = (int)((tail - size) % ring.cap);
  st->ring_size = (int)size;
}
static __global__ __launch_bounds__(256) void k_env_commit(EnvDev e, Ring ring, DevState* st) { env_commit_body(e, ring, st); }

}  // namespace dqnhip
