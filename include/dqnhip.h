/*
 * dqnhip.h — C-ABI of the MI355X-native (gfx950) actor-critic learner.
 *
 * This is the drop-in boundary for ONE hot path of mhauskn/dqn-hfo: the
 * dqn::DQN Update()/SelectAction(s)/AddTransition(s) surface declared in the
 * reference's src/dqn.hpp:56-134.  The reference has no FFI of its own (it is
 * a C++ class linked statically into bin/dqn, CMakeLists.txt:32-33); the
 * entry points below are what a `dqn::DQN` adaptor class binds to (see
 * INTEGRATION.md, include/dqn.hpp and dqn-hfo_amd/csrc/dqn_dropin.cpp).  Every function cites
 * the reference method it replaces.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints, floats.  No torch, no HIP
 *     types in signatures (streams / device memory travel as void*).
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from dqnhip_last_error() (thread-local).  The adaptor turns a
 *     non-zero status into LOG(FATAL), which is the reference's error
 *     convention (glog CHECK/abort, src/dqn.cpp:898,906 ...).
 *   - one handle == one learner == one HIP stream on one device.  Handles are
 *     not thread-safe (the reference uses one DQN per agent thread,
 *     src/dqn_main.cpp:264).
 *   - "host" pointers are ordinary CPU memory; "_device" variants take
 *     pointers to HBM on the handle's device.
 *   - ActorOutput layout (src/dqn.hpp:28, src/dqn.cpp:210-216): 10 floats
 *     [dash, turn, tackle, kick | dashPow, dashAng, turnAng, tackleAng,
 *      kickPow, kickAng].
 *   - dense parameter order (Caffe learnable_params order):
 *       actor : ip1.W[h1,S] ip1.b[h1] ... ip4.W ip4.b
 *               action_layer.W[4,h4] .b[4]  actionpara_layer.W[6,h4] .b[6]
 *       critic: ip1.W[h1,S+10] ip1.b ... ip4.W ip4.b  q_values_layer.W[1,h4] .b[1]
 *     all weights row-major [num_output, K] as in Caffe InnerProduct.
 */
#ifndef DQNHIP_H_
#define DQNHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQNHIP_ACTION_SIZE 4        /* kActionSize       src/dqn.hpp:20 */
#define DQNHIP_ACTION_PARAM_SIZE 6  /* kActionParamSize  src/dqn.hpp:21 */
#define DQNHIP_ACTOR_OUT 10         /* ActorOutput       src/dqn.hpp:28 */
#define DQNHIP_MAX_HIDDEN 8

/* which network a call refers to */
enum dqnhip_net {
  DQNHIP_ACTOR = 0,         /* actor_net_          src/dqn.hpp:189 */
  DQNHIP_CRITIC = 1,        /* critic_net_         src/dqn.hpp:191 */
  DQNHIP_ACTOR_TARGET = 2,  /* actor_target_net_   src/dqn.hpp:193 */
  DQNHIP_CRITIC_TARGET = 3  /* critic_target_net_  src/dqn.hpp:192 */
};

/* which per-parameter array: data, or Adam history (Caffe SolverState
 * history = [m_0..m_P-1, v_0..v_P-1]), or last gradient */
enum dqnhip_param_kind {
  DQNHIP_KIND_W = 0,
  DQNHIP_KIND_M = 1,
  DQNHIP_KIND_V = 2,
  DQNHIP_KIND_G = 3
};

/* Learner configuration.  Defaults in comments are the reference's.
 * Replaces: the compile-time constants of src/dqn.hpp:18-21, the Tower()
 * size list of src/dqn.cpp:425,449, the 11 learner gflags of
 * src/dqn.cpp:21-31 and the solver fields set in src/dqn_main.cpp:249-262. */
#define DQNHIP_FP32 0
#define DQNHIP_FP16 1

typedef struct dqnhip_config {
  int32_t struct_size;       /* = sizeof(dqnhip_config); ABI check            */
  int32_t minibatch;         /* kMinibatchSize (32); multiple of 32           */
  int32_t state_size;        /* num_features = 50 + 9*players; >= 1           */
  int32_t num_hidden;        /* tower depth (4); 1..DQNHIP_MAX_HIDDEN         */
  int32_t hidden[DQNHIP_MAX_HIDDEN]; /* {1024,512,256,128}; multiples of 64   */
  int32_t replay_capacity;   /* FLAGS_memory (500000)                         */
  int32_t soft_update_freq;  /* FLAGS_soft_update_freq (1)                    */
  double gamma;              /* FLAGS_gamma (.99) — double, as the reference  */
  double beta;               /* FLAGS_beta  (.5)                              */
  double tau;                /* FLAGS_tau   (.001); applied as float          */
  float actor_lr;            /* FLAGS_actor_lr  (1e-5)                        */
  float critic_lr;           /* FLAGS_critic_lr (1e-3)                        */
  float momentum;            /* Adam beta1, FLAGS_momentum  (.95)             */
  float momentum2;           /* Adam beta2, FLAGS_momentum2 (.999)            */
  float delta;               /* Adam eps, Caffe SolverParameter.delta (1e-8)  */
  float clip_gradients;      /* FLAGS_clip_grad (10); < 0 disables            */
  int32_t device;            /* HIP device ordinal                            */
  int32_t dp_world;          /* data-parallel world size (1 = single GPU)     */
  int32_t dp_rank;           /* this learner's rank in the DP group           */
  int32_t use_graph;         /* 1: replay the update as a captured hipGraph   */
  uint64_t seed;             /* counter-based RNG key for on-device sampling  */
  void* stream;              /* optional hipStream_t to run on (NULL: own)    */
  void* grad_arena;          /* optional caller-owned device memory that will
                                hold both gradient arenas (so torch.distributed
                                can all-reduce it in place); NULL: library
                                allocates.  Size: dqnhip_grad_arena_bytes().  */
  size_t grad_arena_bytes;
  int32_t precision;         /* DQNHIP_FP32 (default): exact-fp32 MFMA, the parity path.
                                DQNHIP_FP16: tower GEMMs take fp16 operands with fp32
                                accumulation (BASELINE.json config #5); master weights,
                                Adam, heads, TD target and losses stay fp32.  Needs
                                minibatch % 128 == 0 and hidden[i] % 128 == 0.       */
  float loss_scale;          /* FP16 only: multiplies the built-in static scales of the
                                back-propagated gradients (0 or 1: defaults)           */
  int32_t tuning_flags;      /* DQNHIP_TUNE_* bits: A/B switches that select an alternative
                                SCHEDULE of the same arithmetic (0 = the measured winners).
                                The library reads no environment variable; every bit has a
                                parity test (tests/test_gpu_tuning_flags.py).          */
} dqnhip_config;

/* fp16 learner: one wgrad launch per layer (a layer's dgrad + wgrad sharing a launch at small
 * minibatches) instead of ALL wgrads of a net + the bias-gradient sums in one launch. */
#define DQNHIP_TUNE_FP16_WGRAD_PER_LAYER 1
/* fp32 learner: the seed of the critic's dq = -1 backward pass (src/dqn.cpp:918-923) and q(s, mu(s)) from a head-backward
 * launch of their own instead of the top layer's forward epilogue / rider blocks of the chain's last launch. */
#define DQNHIP_TUNE_SEPARATE_HEAD_SEED 2
/* fp32 learner: a tower's backward as wgrad(i) + dgrad(i) per layer and a last launch with the first layer's wgrad alone,
 * instead of the shifted schedule dgrad(L-1) | wgrad(i+1) + dgrad(i) ... | wgrad(1) + wgrad(0) (same launch count). */
#define DQNHIP_TUNE_BWD_UNSHIFTED 4
/* The critic's first-layer action-column input gradient in a launch of its own (+ the q riders) and the inverting gradients + actor
 * heads' backward in another (k_head_bwd<10>), instead of all three in one launch (k_dqda_head_bwd, round 5; any tower-top width and
 * the fp16 learner below 1024 rows since round 6 — there the separate form is the fp16-MFMA layer-0 dgrad launch, and the two agree
 * to fp32 round-off instead of bit for bit). */
#define DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD 8
/* fp32 learner: Step(1)'s head arithmetic (q', q, TD target, loss, dq, the tower-top gradient) in a launch of its own
 * (k_head_q_train) instead of inside the critic's top-layer dgrad launch (k_dgrad_qtrain, round 5: the two head dot products
 * arrive in 16-column pieces from the critics' top forward layers).  The two forms differ by fp32 round-off only (another fixed
 * summation order for q', q; the per-row scalar dq applied after the dgrad's reduction instead of before). */
#define DQNHIP_TUNE_SEPARATE_Q_TRAIN 16
/* fp32 learner: the first tower layer of critic(s, mu(s)) in a launch of its own instead of inside the critic's optimiser launch
 * (FirstLayerRider, round 5: the optimiser workgroups that own W1 run the layer on the weights they have just stepped).  Same bits. */
#define DQNHIP_TUNE_SEPARATE_FIRST_LAYER 32
/* fp32 learner, Step(1): the two critics' first tower layers in a launch of their own behind the actor heads, instead of
 * critic(s, a)'s in the update's first GEMM launch and critic_target(s', mu'(s'))'s split into a state half (that launch) and an
 * action half applied by the target actor's head kernel (round 5).  critic_target's first layer then differs by fp32 round-off
 * (another summation order); everything else is the same arithmetic. */
#define DQNHIP_TUNE_SEPARATE_CRITIC_FIRST_LAYERS 64
/* fp32 learner, multi-update graphs (dqnhip_update_async_n): the next update's gather in the update's LAST launch and its four first
 * layers in a launch of their own (rounds 3-5), instead of the gather in the critic's optimiser launch and the first layers as riders
 * of the actor's optimiser launch (round 5).  Same bits. */
#define DQNHIP_TUNE_LATE_GATHER 128

typedef struct dqnhip_learner* dqnhip_handle;

/* Fill *cfg with the reference defaults (B=32, tower 1024-512-256-128, replay
 * 500k, gamma .99, beta .5, tau .001, Adam .95/.999, lr 1e-5/1e-3, clip 10). */
void dqnhip_default_config(dqnhip_config* cfg, int32_t state_size);

/* Bytes the caller must provide in cfg->grad_arena (0 on invalid config). */
size_t dqnhip_grad_arena_bytes(const dqnhip_config* cfg);

/* Thread-local message of the last failure in this thread. */
const char* dqnhip_last_error(void);

/* Replaces DQN::DQN + DQN::Initialize (src/dqn.cpp:457-483, 622-662):
 * allocates nets (gaussian(0.01) weights, zero biases, src/dqn.cpp:350-352),
 * Adam state, target nets as hard copies (CloneNet, :660-661) and the
 * device-resident replay ring. */
int dqnhip_create(const dqnhip_config* cfg, dqnhip_handle* out);
/* Replaces DQN::~DQN (src/dqn.cpp:485). */
int dqnhip_destroy(dqnhip_handle h);

/* ---- the hot path ------------------------------------------------------ */

/* Replaces DQN::UpdateActorCritic (src/dqn.cpp:828-972): one full update —
 * minibatch gather, target-net forward, TD target, critic Step (fwd, bwd,
 * clip, Adam), actor forward, critic forward, critic input-gradient,
 * inverting gradients, actor backward, clip, Adam, soft target update.
 * idx_host: B logical replay indices in [0, memory_size) — the explicit form
 * of SampleTransitionsFromMemory (src/dqn.cpp:501-509); NULL = sample on the
 * device with the counter-based generator.  Blocks until the update is done
 * and returns (critic_loss, avg_q) exactly as the reference's return value. */
int dqnhip_update(dqnhip_handle h, const int32_t* idx_host,
                  float* critic_loss, float* avg_q);

/* One-deep pipelined dqnhip_update: enqueues update t and returns (critic_loss, avg_q) of update t-1
 * ((0, 0) on the first call).  What dqnhip_update costs beside the kernels — the host's index draw, the
 * H2D copy of the indices and a blocking read-back per update — then overlaps the previous update
 * instead of idling the device; the indices and scalars use two pinned slots each.  Semantic
 * difference to the reference's UpdateActorCritic: the returned pair (and a "Target not finite!" /
 * "Critic loss not finite!" failure) lags by one update; dqnhip_read_stats drains the last one. */
int dqnhip_update_pipelined(dqnhip_handle h, const int32_t* idx_host,
                            float* critic_loss, float* avg_q);

/* dqnhip_update for a caller that runs its updates in bursts of blocking calls — the reference's driver:
 * `for (i < n_updates) dqn->Update()`, src/dqn_main.cpp:359-361, each Update() drawing its indices on the host
 * (src/dqn.cpp:501-509) and returning (critic_loss, avg_q) — and can say which indices its NEXT call will bring
 * (idx_next, NULL: unknown).  The next update's minibatch gather then rides in this update's critic optimiser launch and
 * its four first tower layers in the actor's, as inside dqnhip_update_async_n's sixteen-update graphs, instead of heading
 * the next call's chain (~12 us of ~300).  What rode along is used only if the next call's idx_host equals this call's
 * idx_next element for element AND no entry point changed weights, iteration counters or the replay memory in between
 * (AddTransition(s), set_params, Load*, Restore*, CloneNet, sharing, any other update call, an env step); otherwise the
 * next call simply starts a fresh chain.  Every update computes what dqnhip_update computes on the same indices, bit for
 * bit.  Learners the riders do not fit (see dqnhip_get_update_plan: early_gather_l0; fp16, data-parallel, sharing
 * learners; use_graph = 0) run dqnhip_update.  The drop-in draws idx_next from a COPY of its std::mt19937 and adopts the
 * copy's state only when the prediction held, so the engine's observable call order is the reference's.
 * critic_loss == avg_q == NULL: enqueue only (the caller works while the update runs — the drop-in draws the prediction
 * after next — and collects the pair with dqnhip_read_stats). */
int dqnhip_update_chained(dqnhip_handle h, const int32_t* idx_host, const int32_t* idx_next,
                          float* critic_loss, float* avg_q);

/* Same update, enqueued on the stream without a host sync (the scalars stay
 * on the device; read them with dqnhip_read_stats). */
int dqnhip_update_async(dqnhip_handle h, const int32_t* idx_host);

/* n such updates back to back with on-device sampling — the reference's inner loops `for (i < n_updates)
 * dqn->Update()` (src/dqn_main.cpp:359-361) and DQN::Benchmark (src/dqn.cpp:487-498) as ONE call.  Exactly the
 * state n calls of dqnhip_update_async(h, NULL) leave (bit for bit); with use_graph the updates are replayed
 * sixteen to a hipGraph launch: the GPU idles ~8 us between two graph launches, and inside such a graph the minibatch
 * gather and the four first tower layers of update u + 1 ride in update u's two optimiser launches instead of heading the
 * next chain (~12 us; DQNHIP_TUNE_LATE_GATHER: the gather alone, in update u's last launch). */
int dqnhip_update_async_n(dqnhip_handle h, int32_t n);

/* Data-parallel form of the same update, cut at its two exchange points:
 *   phase 0: gather .. critic backward      -> critic gradients ready
 *   phase 1: critic clip+Adam(+soft update), actor fwd, critic fwd, critic
 *            input-gradient, inverting gradients, actor backward
 *                                            -> actor gradients ready
 *   phase 2: actor clip+Adam(+soft update), iteration counters
 * Between phases the caller sum-all-reduces dqnhip_grad_buffer(net) across
 * the DP group (RCCL).  With dp_world == 1 running 0,1,2 back to back is
 * identical to dqnhip_update_async.
 * Overlap form: phase 10 = phase 0 without the online actor's forward mu(s)
 * (src/dqn.cpp:910-911), phase 11 = that forward; 11 touches neither gradient
 * arena, so 10, [start all-reduce of the critic gradients], 11, [wait], 1, .. hides
 * it behind the collective.  10 + 11 compute exactly what 0 computes. */
int dqnhip_update_phase(dqnhip_handle h, int32_t phase, const int32_t* idx_host);
/* Abandons a phased update that will not be completed (the caller's exchange step failed): the
 * next dqnhip_update_phase / dqnhip_update* starts a fresh update.  A phase that itself returns
 * non-zero abandons the update on its own.  Weights already stepped by phase 1 stay stepped. */
int dqnhip_update_abort(dqnhip_handle h);

/* Replaces one solver's ApplyUpdate() in isolation (actor_solver_->ApplyUpdate(), src/dqn.cpp:964; the
 * tail of critic_solver_->Step(1), :904): ClipGradients + Adam + Net::Update on the gradient currently in
 * the net's arena (e.g. written with dqnhip_set_params(.., DQNHIP_KIND_G, ..)), the soft update of that
 * net's target under the reference's condition (:967: max_iter() after both solvers' increments of a full
 * update), then set_iter(iter() + 1) of that solver (:965).  net = DQNHIP_ACTOR or DQNHIP_CRITIC.
 * The optimiser pass of dqnhip_update is this same kernel: tests pin it here on identical (w, g, m, v,
 * w', iter) to a few ulp. */
int dqnhip_apply_update(dqnhip_handle h, int32_t net);
/* The same step evaluated the way a `world`-rank group with a SHARDED optimiser (DQNHIP_DP_SHARD_OPT, below) evaluates it,
 * this one learner standing in for every rank in turn: each slice's share of the clip norm, their sum in rank order (what
 * the group's 4-float all-reduce leaves on every rank), clip + Adam + soft update slice by slice with that norm.  The
 * gradient in the arena already is the reduced one, so no exchange is involved: this pins the slice arithmetic and the
 * norm's composition on ONE GPU.  world = 1 equals dqnhip_apply_update bit for bit; world > 1 differs only through the
 * summation order of the norm (identical bits whenever the clip is inactive). */
int dqnhip_apply_update_sharded(dqnhip_handle h, int32_t net, int32_t world);

/* ---- the launch plan of this learner's update ---------------------------------------------
 * No reference counterpart (Caffe runs one layer at a time).  Which merged forms of the launch sequence behind
 * dqnhip_update* this learner takes — they depend on its shapes, its tuning flags, sharing and the data-parallel mode —
 * and how many kernels one update launches, COUNTED from a stream capture of the very sequence dqnhip_update* enqueues
 * (nothing executes).  Exists so that a shape predicate that stops matching shows up as a failed assertion
 * (tests/test_gpu_update_plan.py; bench.py prints it in `config.plan`) instead of as a silently slower schedule. */
#define DQNHIP_PLAN_FP16 1                      /* cfg.precision == DQNHIP_FP16 */
#define DQNHIP_PLAN_DATA_PARALLEL 2             /* phases cut for an exchange: dp_world > 1, or a one-rank group with bf16 exchange / sharded optimiser */
#define DQNHIP_PLAN_BWD_SHIFTED_CRITIC 4        /* the shifted backward schedule (see DQNHIP_TUNE_BWD_UNSHIFTED) in Step(1) */
#define DQNHIP_PLAN_BWD_SHIFTED_ACTOR 8         /* ... in the actor's backward */
#define DQNHIP_PLAN_HEAD_WGRAD_RIDES_CRITIC 16  /* the critic head's dW / db as rider blocks of the net's last backward launch (both precisions) */
#define DQNHIP_PLAN_HEAD_WGRAD_RIDES_ACTOR 32   /* ... the actor heads' */
#define DQNHIP_PLAN_Q_TRAIN_IN_DGRAD 64         /* fp32: k_dgrad_qtrain (see DQNHIP_TUNE_SEPARATE_Q_TRAIN); fp16: k_head_q_train also writes the tower-top gradient — either way Step(1) has no head-backward launch */
#define DQNHIP_PLAN_HEAD_SEED_FUSED 128         /* the dq = -1 seed from the top layer's forward epilogue (DQNHIP_TUNE_SEPARATE_HEAD_SEED) */
#define DQNHIP_PLAN_DQDA_HEAD_BWD 256           /* k_dqda_head_bwd (DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD) */
#define DQNHIP_PLAN_CRITIC_L0_RIDES 512         /* critic(s, mu(s))'s first layer inside the critic's optimiser launch (DQNHIP_TUNE_SEPARATE_FIRST_LAYER) */
#define DQNHIP_PLAN_FIRST_LAYERS_MERGED 1024    /* Step(1)'s four first layers in one launch (DQNHIP_TUNE_SEPARATE_CRITIC_FIRST_LAYERS) */
#define DQNHIP_PLAN_EARLY_GATHER_L0 2048        /* multi-update graphs: next gather / next first layers ride in the two optimiser launches (DQNHIP_TUNE_LATE_GATHER) */
#define DQNHIP_PLAN_DP_TAILS_RIDE 4096          /* data parallel: the [loss, q, flag] tails of the exchange ride in each net's last backward launch */
typedef struct dqnhip_update_plan {
  int32_t struct_size;          /* in: sizeof(dqnhip_update_plan) */
  int32_t forms;                /* DQNHIP_PLAN_* bits */
  int32_t launches_single;      /* kernels of one stand-alone update: eager, dqnhip_update, the one-update graph */
  int32_t launches_graph_first; /* ... of the first update of a multi-update graph (dqnhip_update_async_n / dqnhip_dp_update_n) */
  int32_t launches_in_graph;    /* ... of every later update of such a graph */
  int32_t updates_per_graph;    /* 16 */
  int32_t collectives;          /* RCCL calls per update once a communicator is up (not counted in launches_*), else 0 */
  int32_t reserved;
} dqnhip_update_plan;
/* Fails while a phased update is in progress or kernel timing is on; launches_* are 0 for a sharded optimiser (its
 * sequence contains collectives and is not captured here). */
int dqnhip_get_update_plan(dqnhip_handle h, dqnhip_update_plan* out);

/* Device pointer + length (floats) of one net's gradient arena, including a
 * 4-float tail [loss_sum, q_sum, 0, 0] so the two reported scalars ride in the
 * same all-reduce.  net = DQNHIP_ACTOR or DQNHIP_CRITIC. */
int dqnhip_grad_buffer(dqnhip_handle h, int32_t net, void** dptr, size_t* nfloats);

/* ---- native data parallelism: RCCL over xGMI inside the library --------------
 * The reference scales with threads + one mutex (src/dqn_main.cpp:62-63, 359-363); there is no
 * collective to replace.  One learner per GPU (cfg.dp_world / cfg.dp_rank), each with its own
 * replay shard; dqnhip_dp_update runs phase 0, sum-all-reduces the critic gradient arena, phase 1,
 * all-reduces the actor's, phase 2 — all on the learner's stream, no host sync.
 *   dqnhip_dp_unique_id   rank 0 creates the RCCL id (DQNHIP_DP_ID_BYTES); the launcher ships it
 *   dqnhip_dp_init        every rank: ncclCommInitRank, then rank 0's weights (4 nets), Adam
 *                         history and iterations are broadcast so the replicas start identical
 *   dqnhip_dp_init_file   the same with files as the rendezvous (one node, no launcher support):
 *                         dqnhip_dp_rendezvous_file + dqnhip_dp_init; rank 0 removes the files once the group is up
 *   dqnhip_dp_rendezvous_file   the rendezvous alone (needs no GPU): rank 0 passes the id in, ranks 1..world-1
 *                         receive it.  Every waiter publishes <path>.req<rank> with a fresh nonce and accepts <path>
 *                         only if it carries that nonce, rank 0 clears leftovers first: a file from an earlier or a
 *                         crashed job can never hand out a dead id, and the same path can be reused.
 * flags: 0 — ONE sum all-reduce per net (arena + 4-float tail), the supported form, what bench.py --gpus N and the drop-in use.
 * DQNHIP_DP_HALF_GRADS: the gradient arenas cross the links as bf16 — half the bytes (6.4 instead of 12.9 MB per
 * net at 4x1024); the [loss, q, flag] tails stay fp32 and travel once, with the actor's gradients.  The reduced
 * gradient then carries 8 significant bits (tests/test_gpu_dp_hip.py bounds the effect); meant for the fp16 learner.
 * cfg.use_graph: dqnhip_dp_update captures phase 0 / all-reduce / phase 1 / all-reduce / phase 2 ONCE and replays
 * it as a hipGraph (eager if RCCL refuses the capture: dqnhip_dp_graph_active tells).
 * Since round 6 a data-parallel learner runs the single learner's merged launches (dqnhip_get_update_plan): the riders sit
 * between the exchange points, the tails of the exchange ride in each net's last backward launch.
 * (Two further exchange forms exist behind DQNHIP_DP_UNVERIFIED_OK — see the end of this header.) */
#define DQNHIP_DP_ID_BYTES 128
#define DQNHIP_DP_PER_LAYER 1
#define DQNHIP_DP_HALF_GRADS 2
#define DQNHIP_DP_SHARD_OPT 4
#define DQNHIP_DP_UNVERIFIED_OK 256     /* lets DQNHIP_DP_PER_LAYER / DQNHIP_DP_SHARD_OPT through for dp_world > 1 (see the end of this header) */
/* ncclGetVersion() and the shared object RCCL was resolved from in this process (a PyTorch host process resolves torch's
 * bundled librccl, a bare host /opt/rocm's); dqnhip_dp_init cross-checks the version over the group and fails on a mix. */
int dqnhip_dp_info(int32_t* rccl_version, char* path, size_t path_bytes);
int dqnhip_dp_unique_id(void* id_out, size_t bytes);
int dqnhip_dp_init(dqnhip_handle h, const void* id, size_t bytes, int32_t flags);
int dqnhip_dp_init_file(dqnhip_handle h, const char* path, int32_t flags, int32_t timeout_s);
int dqnhip_dp_rendezvous_file(const char* path, int32_t rank, int32_t world, int32_t timeout_s, void* id, size_t bytes);
int dqnhip_dp_rendezvous_cleanup(const char* path, int32_t world);
int dqnhip_dp_graph_active(dqnhip_handle h, int32_t* active);
int dqnhip_dp_broadcast_params(dqnhip_handle h, int32_t root);
int dqnhip_dp_update(dqnhip_handle h, const int32_t* idx_host);
/* n of them with on-device sampling: dqnhip_update_async_n for a group (every rank calls it with the same n; sixteen
 * updates, collectives included, per hipGraph launch). */
int dqnhip_dp_update_n(dqnhip_handle h, int32_t n);
int dqnhip_dp_gather_state(dqnhip_handle h);
int dqnhip_dp_destroy(dqnhip_handle h);

/* Waits for the stream and returns the scalars of the last update.  Fails — the reference
 * aborts: CHECK(std::isfinite(target)) src/dqn.cpp:898, CHECK(std::isfinite(critic_loss)) :906 —
 * with "Target not finite!" / "Critic loss not finite!" if any update since the last call
 * (blocking, async, phased or graph-replayed alike: the flags are raised on the device and are
 * sticky until reported here) produced a non-finite TD target or loss, or if a clip+Adam step was
 * skipped because the gradient norm was not finite (fp16 overflow; weights are left untouched). */
int dqnhip_read_stats(dqnhip_handle h, float* critic_loss, float* avg_q);
/* Optimiser steps skipped so far because of a non-finite gradient norm (one count per net per update).
 * A skipped step still ends the update normally: the iteration counters, the soft-update schedule and
 * the sampling counter advance (as they would after Caffe's Step with a zero update); only w, m, v and
 * the targets are left untouched.  Under native data parallelism every rank takes the same decision (the
 * norm is that of the REDUCED gradient) and a rank-local "Target not finite!" is shared through the
 * all-reduced tail, so all ranks report the same error at the same update. */
int dqnhip_skipped_steps(dqnhip_handle h, int64_t* count);

/* Sum-reduce the gradient arenas (tails included) of n learners of one data-parallel group that
 * live on ONE device, in rank order, leaving the sum in each: the exchange step between
 * dqnhip_update_phase calls without a communicator (co-located agents; the one-GPU test of the
 * dp_world > 1 arithmetic).  Groups that span devices use dqnhip_dp_* (RCCL over xGMI). */
int dqnhip_reduce_gradients_local(dqnhip_handle* learners, int32_t n, int32_t net);

/* Replaces DQN::Benchmark (src/dqn.cpp:487-498): times `iterations` updates
 * (after `warmup` untimed ones) with HIP events on the learner's stream and
 * returns the average in milliseconds ("Average Update: X ms"). */
int dqnhip_benchmark(dqnhip_handle h, int32_t warmup, int32_t iterations,
                     float* avg_ms);

/* DQN::Benchmark as the reference's driver experiences it through the drop-in (src/dqn.cpp:487-498 loops over
 * UpdateActorCritic(): indices drawn on the host with std::mt19937 + uniform_int_distribution, :501-509, and a
 * blocking (critic_loss, avg_q) per update): wall-clock average over `iterations` calls of dqnhip_update
 * (pipelined = 0), dqnhip_update_pipelined (pipelined = 1) or dqnhip_update_chained with the next call's indices drawn one
 * call ahead (pipelined = 2: what the drop-in's UpdateActorCritic() does) after `warmup` untimed ones. */
int dqnhip_benchmark_blocking(dqnhip_handle h, int32_t warmup, int32_t iterations, uint64_t seed,
                              int32_t pipelined, float* avg_ms);

/* Replaces the greedy branch of DQN::SelectActions -> SelectActionGreedily
 * (src/dqn.cpp:695-711, 734-766): actor forward on n states [n, S] (host),
 * returns [n, 10] ActorOutputs (host).  n is not capped at kMinibatchSize.
 * The epsilon draw and GetRandomActorOutput stay in the caller so the
 * reference's std::mt19937 call order is preserved (src/dqn.cpp:700). */
int dqnhip_select_actions(dqnhip_handle h, const float* states_host, int32_t n,
                          float* actor_out_host);
/* Same with states / outputs in HBM ([n,S] and [n,10], dense). */
int dqnhip_select_actions_device(dqnhip_handle h, const float* states_dev,
                                 int32_t n, float* actor_out_dev);
/* Target-actor variant (SelectActionGreedily(*actor_target_net_, ...)). */
int dqnhip_select_actions_net(dqnhip_handle h, int32_t net,
                              const float* states_host, int32_t n,
                              float* actor_out_host);

/* Replaces DQN::CriticForward / EvaluateAction (src/dqn.cpp:982-1020,
 * 688-693): Q(s, a) for n host rows using `net` (CRITIC or CRITIC_TARGET). */
int dqnhip_critic_forward(dqnhip_handle h, int32_t net, const float* states_host,
                          const float* actor_out_host, int32_t n, float* q_host);

/* Replaces DQN::AddTransitions (src/dqn.cpp:775-781): FIFO-evicts while
 * size + n >= capacity, then appends n transitions.  terminal[i] != 0 means
 * next_state == boost::none (src/dqn.cpp:878); next_states rows of terminal
 * transitions are ignored. */
int dqnhip_add_transitions(dqnhip_handle h, const float* states, const float* actor_out,
                           const float* rewards, const float* on_policy_targets,
                           const float* next_states, const uint8_t* terminal, int32_t n);
/* Replaces DQN::AddTransition (src/dqn.cpp:768-773): evicts one iff size ==
 * capacity, then appends one. */
int dqnhip_add_transition(dqnhip_handle h, const float* state, const float* actor_out,
                          float reward, float on_policy_target,
                          const float* next_state, uint8_t terminal);
/* AddTransitions with all arrays already in HBM (batched env workers). */
int dqnhip_add_transitions_device(dqnhip_handle h, const float* states, const float* actor_out,
                                  const float* rewards, const float* on_policy_targets,
                                  const float* next_states, const uint8_t* terminal, int32_t n);

/* Replaces DQN::LabelTransitions (src/dqn.cpp:783-797) for one episode held in
 * host arrays: on_policy_target[last] = r[last]; [i] = r[i] + gamma*[i+1]
 * (gamma double, result rounded to float). */
int dqnhip_label_transitions(double gamma, const float* rewards, int32_t n,
                             float* on_policy_targets);

/* Replaces DQN::SampleStatesFromMemory (src/dqn.cpp:511-523): the states ([n,S], host) of n
 * transitions sampled uniformly with replacement.  idx_host: explicit logical indices (the
 * SampleTransitionsFromMemory part, src/dqn.cpp:501-509) or NULL = counter-based draw on the device. */
int dqnhip_sample_states(dqnhip_handle h, const int32_t* idx_host, int32_t n, float* states_host);
/* Replaces getActorOutput (src/dqn.cpp:719-732): the first batch_size rows [batch_size,10] of the
 * output blobs an actor's last minibatch forward left behind (ACTOR: mu(s) of the last update,
 * ACTOR_TARGET: mu'(s')). */
int dqnhip_get_actor_output(dqnhip_handle h, int32_t net, int32_t batch_size, float* actor_out_host);

/* DQN::memory_size / ClearReplayMemory (src/dqn.hpp:106,112). */
int dqnhip_memory_size(dqnhip_handle h, int32_t* size);
int dqnhip_clear_memory(dqnhip_handle h);
/* Read logical transitions [first, first+n) back to host arrays (for
 * DQN::SnapshotReplayMemory, src/dqn.cpp:1146-1178).  Any pointer may be NULL. */
int dqnhip_read_memory(dqnhip_handle h, int32_t first, int32_t n, float* states,
                       float* actor_out, float* rewards, float* on_policy_targets,
                       float* next_states, uint8_t* terminal);

/* Replaces DQN::SnapshotReplayMemory / LoadReplayMemory (src/dqn.cpp:1146-1226): the
 * reference's `.replaymemory` file — a gzip stream holding int32 num_transitions, then per
 * transition: state[S] floats, ActorOutput[10] floats, float reward, float on_policy_target,
 * 1-byte bool terminal (= next_state is none).  Next states are not stored: transition i's
 * next state is transition i+1's state when i is not terminal; a trailing non-terminal
 * transition loads as terminal (boost::none), exactly as in the reference.  Load clears the
 * memory first and does not evict (the reference resizes the deque to num_transitions);
 * it fails if num_transitions exceeds the capacity. */
int dqnhip_snapshot_replay_memory(dqnhip_handle h, const char* filename);
int dqnhip_load_replay_memory(dqnhip_handle h, const char* filename);

/* ---- parameters, iterations, targets ----------------------------------- */

/* number of learnable parameters of a net in the dense order above */
int dqnhip_param_count(dqnhip_handle h, int32_t net, size_t* count);
/* Copy a whole net's dense parameter vector out / in.  Used by Snapshot /
 * Restore / LoadWeights (src/dqn.cpp:525-557, 586-620).  kind M/V/G only for
 * ACTOR and CRITIC. */
int dqnhip_get_params(dqnhip_handle h, int32_t net, int32_t kind, float* host, size_t count);
int dqnhip_set_params(dqnhip_handle h, int32_t net, int32_t kind, const float* host, size_t count);
/* CloneNet (src/dqn.cpp:1022-1035): hard copy online -> target.
 * net = DQNHIP_ACTOR or DQNHIP_CRITIC (the source). */
int dqnhip_clone_to_target(dqnhip_handle h, int32_t net);
/* Solver::iter / set_iter (src/dqn.hpp:129-130, src/dqn.cpp:965). */
int dqnhip_get_iters(dqnhip_handle h, int32_t* actor_iter, int32_t* critic_iter);
int dqnhip_set_iters(dqnhip_handle h, int32_t actor_iter, int32_t critic_iter);

/* ---- multi-agent sharing (src/dqn.cpp:1036-1083; src/dqn_main.cpp:305-323) ------------
 * DQN::ShareParameters(other, n_actor, n_critic): the first n layers-with-blobs (Caffe layer
 * order: ip1..ipL, then the head layers) of `other`'s actor / critic AND of its two target nets
 * read and write `owner`'s storage from now on (Blob::ShareData — data only: gradients and
 * Adam history stay per learner, exactly the reference's Hogwild arrangement; both learners'
 * solvers and soft updates write the shared weights).  n may be 0..L (tower layers) or all
 * layers; splitting the actor's two heads is refused.  Both learners must be on one device
 * and of the same shape.  Calling again changes the layer counts; (0, 0) un-shares.
 * `owner` must outlive `other` (dqnhip_destroy(owner) fails while sharers exist). */
int dqnhip_share_parameters(dqnhip_handle owner, dqnhip_handle other, int32_t num_actor_layers,
                            int32_t num_critic_layers);
/* DQN::ShareReplayMemory(other) (src/dqn.cpp:1080-1082): `other` drops its own deque and uses
 * `owner`'s ring (capacity included) for AddTransition(s), sampling, memory_size, snapshot
 * and load.  Users of a shared ring on different streams are ordered in host-call order by
 * events; a private ring pays nothing. */
int dqnhip_share_replay_memory(dqnhip_handle owner, dqnhip_handle other);

/* The configuration the learner was created with (for clients that need the shapes). */
int dqnhip_get_config(dqnhip_handle h, dqnhip_config* out);

/* ---- Caffe snapshot layout (src/dqn.cpp:525-620; Caffe Solver::Snapshot/Restore) -------
 * `.caffemodel` = binary caffe.NetParameter{name=1, layer=100{name=1, type=2, blobs=7}},
 * `.solverstate` = binary caffe.SolverState{iter=1, learned_net=2, history=3, current_step=4},
 * blobs = caffe.BlobProto{shape=7{dim=1 packed}, data=5 packed float}.  Layer names are the
 * reference's: ip<i>_layer (src/dqn.cpp:406), action_layer, actionpara_layer (:426-427),
 * q_values_layer (src/dqn.hpp:43); weight blob [num_output, K] then bias [num_output]; Adam
 * history = [m of every param in order, then v of every param].
 * net = DQNHIP_ACTOR or DQNHIP_CRITIC. */
/* Net::ToProto + WriteProtoToBinaryFile */
int dqnhip_save_caffemodel(dqnhip_handle h, int32_t net, const char* filename);
/* Net::CopyTrainedLayersFrom (layers matched by NAME, others ignored) followed by CloneNet to
 * the target net — DQN::LoadActorWeights / LoadCriticWeights (src/dqn.cpp:525-539) */
int dqnhip_load_caffemodel(dqnhip_handle h, int32_t net, const char* filename);
/* Solver::Snapshot: writes <prefix>_iter_<iter>.caffemodel and .solverstate (learned_net =
 * the caffemodel's path), returns the iteration */
int dqnhip_solver_snapshot(dqnhip_handle h, int32_t net, const char* prefix, int32_t* iter_out);
/* Solver::Restore + CloneNet — DQN::RestoreActorSolver / RestoreCriticSolver (:541-557):
 * iteration, learned_net weights, Adam history; the target net becomes a hard copy */
int dqnhip_solver_restore(dqnhip_handle h, int32_t net, const char* solverstate);
/* DQN::Snapshot (src/dqn.cpp:586-620): both solvers under save_path, renamed to
 * snapshot_prefix, optional `<prefix>_iter_<max_iter>.replaymemory`, optional removal of
 * older snapshots of the same prefix */
int dqnhip_snapshot(dqnhip_handle h, const char* save_path, const char* snapshot_prefix,
                    int32_t remove_old, int32_t snapshot_memory);
/* FindLatestSnapshot (src/dqn.cpp:122-144): newest <prefix>_{actor,critic}_iter_N.solverstate
 * and <prefix>_iter_N.replaymemory; each buffer receives "" when none is found */
int dqnhip_find_latest_snapshot(const char* snapshot_prefix, char* actor, char* critic, char* memory,
                                size_t buf_len);
/* FindHiScore (src/dqn.cpp:146-158) */
int dqnhip_find_hiscore(const char* snapshot_prefix, int32_t* score);
/* RemoveFilesMatchingRegexp (src/dqn.cpp:92-98) */
int dqnhip_remove_files_matching_regexp(const char* regexp);
/* FilesMatchingRegexp (src/dqn.hpp:213-216; src/dqn.cpp:559-580): regular files in the regexp's
 * directory whose file NAME matches its last path component; paths '\n'-separated (sorted) in
 * buf, their number in *count.  Fails if buf is too small. */
int dqnhip_files_matching_regexp(const char* regexp, char* buf, size_t buf_len, int32_t* count);
/* RemoveSnapshots (src/dqn.hpp:222-224; src/dqn.cpp:100-109): removes the matching files whose
 * iteration (the number between the last '_' and the last '.') is < min_iter. */
int dqnhip_remove_snapshots(const char* regexp, int32_t min_iter);

/* ---- introspection for parity tests ------------------------------------ */

/* Copy a named [B, *] intermediate of the last update to the host:
 * "q_target" [B] (Q'(s',mu'(s'))), "y" [B] (TD target), "q_train" [B],
 * "q_policy" [B], "actor_out" [B,10] (mu(s)), "dq_da" [B,10] (post
 * inverting-gradients), "idx" [B] (as float), "terminal" [B].
 * count = floats the caller's buffer holds; fails if too small. */
int dqnhip_debug_read(dqnhip_handle h, const char* name, float* host, size_t count);

/* The HIP stream (hipStream_t) the learner enqueues on. */
int dqnhip_get_stream(dqnhip_handle h, void** stream);
/* Average duration (ms) of the dominant kernel family over the launches since
 * the last call with reset != 0; measured with HIP events when profiling is
 * enabled via dqnhip_set_kernel_timing.  Used by bench.py's roofline leg. */
int dqnhip_set_kernel_timing(dqnhip_handle h, int32_t enable);
int dqnhip_get_kernel_timing(dqnhip_handle h, const char* family, float* avg_ms,
                             int64_t* launches, int32_t reset);

/* ---- data parallelism: exchange forms that are NOT part of the supported surface -----------------------------------------
 * Built in rounds 3-4, priced net-negative by this library's own model and never run on more than one rank (no box
 * with more than one GPU has been available): dqnhip_dp_init refuses them for dp_world > 1 unless DQNHIP_DP_UNVERIFIED_OK says the
 * caller knows.  One-rank groups stay open (tests pin their launch sequences).
 * DQNHIP_DP_PER_LAYER buckets each all-reduce per tower layer on a communication stream, started as soon as that layer's backward
 * launch has run (backward order), head + tail last (fp32 learner only).
 * DQNHIP_DP_SHARD_OPT: the optimiser sharded over the group (ZeRO-1 style) instead of replicated: per net, a reduce-scatter leaves
 * rank r with slice r of the summed gradient, the ranks all-reduce a 4-float tail {loss, q, target flag, sum of squares of their
 * slice} — the clip norm — each rank runs clip + Adam + soft update on its 1/N slice only, and the updated online AND target weights
 * (the targets move every update, src/dqn.cpp:967-970; fp16 learner: the fp16 mirrors too) are all-gathered.  m and v of foreign
 * slices go stale: dqnhip_dp_gather_state (collective) brings them back before a snapshot / dqnhip_get_params(KIND_M, KIND_V);
 * dqnhip_dp_destroy refuses until it has been called since the last update.  Combines with DQNHIP_DP_HALF_GRADS, not with
 * DQNHIP_DP_PER_LAYER; the arena must be divisible by 4 x dp_world.  dqnhip_apply_update_sharded pins its slice arithmetic on one GPU. */

/* ---- batched env front-end (HFOGameState reward shaping, N workers) ----- */

/* Replaces HFOGameState::update + reward (src/hfo_game.cpp:122-236) for n
 * synthetic workers whose state vectors are in HBM: see dqnhip_env.h. */

#ifdef __cplusplus
}
#endif
#endif /* DQNHIP_H_ */
