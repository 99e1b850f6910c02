/*
 * dqn_oracle.c — CPU restatement of the reference's hot path.  TEST
 * INFRASTRUCTURE ONLY: nothing in the product path (dqn-hfo_amd/) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED.  The arithmetic of the reference's hot path lives in an
 * un-vendored third-party dependency — BVLC/caffe @
 * 2ef584785c8ade90260eb117f189146364494183 (reference README.md:7-39) — which
 * is absent from /root/reference and from this image, and the reference ships
 * no tests, golden vectors or fixtures.  This file
 * restates Caffe's published algorithms for the layers/solver the reference
 * reaches (InnerProduct, ReLU(negative_slope), Concat, EuclideanLoss,
 * SGDSolver::ClipGradients, AdamSolver::ComputeUpdateValue, Net::Update) and
 * follows the reference's own call sequence line by line.  It is cross-checked
 * by an independent float64/autograd restatement (oracle/torch_ref.py) and by
 * hand-derived known-answer tests (tests/test_oracle_kat.py).
 *
 * Numerics: fp32 storage everywhere (as the reference); every GEMM dot product
 * (Caffe: cblas_sgemm, whose fp32 summation order is BLAS-implementation-
 * defined and unknowable here) is accumulated in double over k and rounded to
 * float ONCE, i.e. it is the correctly rounded value every fp32 summation order
 * scatters around by a few ulp — an order-free definition of the sgemm result.
 * (Rounds 1-3 used a k-ordered fp32 fmaf chain: one arbitrary order among many,
 * and at 4 x 1024 units its own round-off flipped the ReLU' of a near-zero
 * pre-activation about once per update relative to float64, which forced a
 * 5e-3 tolerance onto a comparison that is 1e-7 otherwise.)  The bias is added
 * last, in float (Caffe does GEMM with beta=0, then a rank-1 bias GEMM:
 * inner_product_layer.cpp Forward_cpu); the optimiser, the TD target and the
 * loss are float exactly where the reference's are, doubles only where the
 * reference uses doubles (src/dqn.cpp:791, 894-897, 915).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXL 8
#define ORC_NA 4   /* kActionSize      src/dqn.hpp:20 */
#define ORC_NP 6   /* kActionParamSize src/dqn.hpp:21 */
#define ORC_NO 10  /* ActorOutput      src/dqn.hpp:28 */

typedef struct {
  int32_t B, S, L;
  int32_t hidden[ORC_MAXL];
  int32_t capacity;          /* FLAGS_memory src/dqn.cpp:25 */
  int32_t soft_update_freq;  /* src/dqn.cpp:23 */
  int32_t global_B;          /* EuclideanLoss normaliser; = B unless data-parallel */
  int32_t mirror_waste;      /* 1: also execute the work the reference computes and
                                discards (critic wgrad in the actor step, first-layer
                                input gradients) — used for the CPU baseline timing */
  double gamma, beta;        /* src/dqn.cpp:24,31 (doubles) */
  float tau;                 /* src/dqn.cpp:22, passed as float :968-969 */
  float lr_actor, lr_critic, beta1, beta2, eps, clip;
} orc_config;

typedef struct {
  int L, in_dim, n_heads;
  int dims[ORC_MAXL + 1];     /* dims[0] = in_dim, dims[l] = hidden[l-1] */
  int head_out[2];            /* actor {4,6}; critic {1} */
  size_t w_off[ORC_MAXL + 2], b_off[ORC_MAXL + 2]; /* tower layers 0..L-1, heads L.. */
  size_t count;
} orc_layout;

typedef struct {
  orc_config cfg;
  orc_layout la, lc;
  /* dense parameter vectors, Caffe learnable_params order (weight, bias per layer) */
  float *w[4];                /* actor, critic, actor_target, critic_target */
  float *g[2], *m[2], *v[2];  /* actor, critic */
  int iter[2];                /* actor_iter, critic_iter */
  /* replay ring modelling std::deque<Transition> (src/dqn.hpp:187) */
  float *r_state, *r_act, *r_rew, *r_mc, *r_next;
  uint8_t *r_term;
  int64_t head, size;
  /* minibatch + intermediates of the last update (debug / parity) */
  float *mb_s, *mb_a, *mb_r, *mb_mc, *mb_n, *q_target, *y, *q_train, *q_policy,
        *actor_out, *dq_da;
  uint8_t *mb_t;
  int32_t *mb_idx;
  /* activations */
  float *actA[ORC_MAXL + 1], *actC[ORC_MAXL + 1], *dA[ORC_MAXL + 1], *dC[ORC_MAXL + 1];
  float last_loss, last_avgq;
  float tail[2][4];           /* per-net [loss_sum, q_sum, 0, 0] for DP all-reduce */
} orc;

/* number of OpenMP threads used by the dense kernels (CPU-baseline timing) */
void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ layout */

static void layout_init(orc_layout *l, int in_dim, const orc_config *c, int actor) {
  l->L = c->L; l->in_dim = in_dim; l->dims[0] = in_dim;
  for (int i = 0; i < c->L; ++i) l->dims[i + 1] = c->hidden[i];
  size_t off = 0;
  for (int i = 0; i < c->L; ++i) {          /* Tower(), src/dqn.cpp:400-416 */
    l->w_off[i] = off; off += (size_t)l->dims[i + 1] * l->dims[i];
    l->b_off[i] = off; off += l->dims[i + 1];
  }
  if (actor) {                               /* src/dqn.cpp:426-427 */
    l->n_heads = 2; l->head_out[0] = ORC_NA; l->head_out[1] = ORC_NP;
  } else {                                   /* src/dqn.cpp:450 */
    l->n_heads = 1; l->head_out[0] = 1; l->head_out[1] = 0;
  }
  for (int h = 0; h < l->n_heads; ++h) {
    l->w_off[c->L + h] = off; off += (size_t)l->head_out[h] * l->dims[c->L];
    l->b_off[c->L + h] = off; off += l->head_out[h];
  }
  l->count = off;
}

/* ------------------------------------------------------------ dense kernels */

/* Caffe ReLULayer::Forward_cpu with negative_slope (src/dqn.cpp:292-301):
 * top = max(x,0) + slope*min(x,0) */
static inline float lrelu(float x) {
  const float slope = 0.01f;
  return fmaxf(x, 0.0f) + slope * fminf(x, 0.0f);
}
/* Caffe ReLULayer::Backward_cpu, in-place so bottom_data is the output:
 * dx = dy * ((y > 0) + slope * (y <= 0)) */
static inline float lrelu_bwd(float dy, float y) {
  const float slope = 0.01f;
  return dy * ((float)(y > 0.0f) + slope * (float)(y <= 0.0f));
}

/* Y[M,N] = act(X[M,K] . W[N,K]^T + b[N])   (InnerProduct forward)
 * double accumulation over k, one rounding to float per output; vectorised across outputs via W^T. */
static void ip_forward(int M, int N, int K, const float *X, int ldx, const float *W,
                       const float *b, float *Y, int ldy, int relu) {
  float *Wt = (float *)malloc((size_t)K * N * sizeof(float));
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
#pragma omp parallel for schedule(static)
  for (int m = 0; m < M; ++m) {
    const float *x = X + (size_t)m * ldx;
    float *y = Y + (size_t)m * ldy;
    for (int n0 = 0; n0 < N; n0 += 64) {
      double acc[64];
      int nb = N - n0 < 64 ? N - n0 : 64;
      for (int j = 0; j < nb; ++j) acc[j] = 0.0;
      for (int k = 0; k < K; ++k) {
        const double xv = (double)x[k];
        const float *wr = Wt + (size_t)k * N + n0;
        for (int j = 0; j < nb; ++j) acc[j] += xv * (double)wr[j];      /* float x float is exact in double */
      }
      for (int j = 0; j < nb; ++j) {
        float t = (float)acc[j] + b[n0 + j];
        y[n0 + j] = relu ? lrelu(t) : t;
      }
    }
  }
  free(Wt);
}

/* dX[M,K] = dY[M,N] . W[N,K]   (InnerProduct backward wrt bottom) */
static void ip_dgrad(int M, int N, int K, const float *dY, int lddy, const float *W,
                     float *dX, int lddx) {
#pragma omp parallel for schedule(static)
  for (int m = 0; m < M; ++m) {
    float *dx = dX + (size_t)m * lddx;
    double *acc = (double *)malloc((size_t)K * sizeof(double));
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int n = 0; n < N; ++n) {
      const double dv = (double)dY[(size_t)m * lddy + n];
      const float *wr = W + (size_t)n * K;
      for (int k = 0; k < K; ++k) acc[k] += dv * (double)wr[k];
    }
    for (int k = 0; k < K; ++k) dx[k] = (float)acc[k];
    free(acc);
  }
}

/* dW[N,K] (+)= dY[M,N]^T . X[M,K];  db[N] (+)= colsum(dY)   (InnerProduct
 * backward wrt params; Caffe accumulates into the diff) */
static void ip_wgrad(int M, int N, int K, const float *dY, int lddy, const float *X,
                     int ldx, float *dW, float *db) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    float *dw = dW + (size_t)n * K;
    double *acc = (double *)malloc((size_t)K * sizeof(double));
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    double bs = 0.0;
    for (int m = 0; m < M; ++m) {
      const double dv = (double)dY[(size_t)m * lddy + n];
      const float *xr = X + (size_t)m * ldx;
      for (int k = 0; k < K; ++k) acc[k] += dv * (double)xr[k];
      bs += dv;
    }
    for (int k = 0; k < K; ++k) dw[k] += (float)acc[k];      /* Caffe accumulates into the diff (which the caller zeroed) */
    db[n] += (float)bs;
    free(acc);
  }
}

/* --------------------------------------------------------------- networks */

/* Tower + heads forward.  acts[0] must hold the input [n, in_dim];
 * acts[l] receives the post-ReLU output of tower layer l (in-place ReLU,
 * src/dqn.cpp:409-410).  head_out: [n, sum(head_out)]. */
static void net_forward(const orc_layout *l, const float *w, float **acts, int n,
                        float *head_out, int ld_head) {
  for (int i = 0; i < l->L; ++i)
    ip_forward(n, l->dims[i + 1], l->dims[i], acts[i], l->dims[i], w + l->w_off[i],
               w + l->b_off[i], acts[i + 1], l->dims[i + 1], 1);
  int col = 0;
  for (int h = 0; h < l->n_heads; ++h) {
    ip_forward(n, l->head_out[h], l->dims[l->L], acts[l->L], l->dims[l->L],
               w + l->w_off[l->L + h], w + l->b_off[l->L + h], head_out + col, ld_head, 0);
    col += l->head_out[h];
  }
}

/* Backward from the head diffs down the tower.
 * d_head [n, sum(head_out)]; dacts[l] scratch [n, dims[l]].
 * want_wgrad: accumulate dW/db into g (Caffe semantics: +=).
 * want_input_grad: also compute d/d(input) into dacts[0] (force_backward,
 * src/dqn.cpp:421,434). */
static void net_backward(const orc_layout *l, const float *w, float *g, float **acts,
                         float **dacts, int n, const float *d_head, int ld_head,
                         int want_wgrad, int want_input_grad) {
  const int L = l->L, H = l->dims[L];
  /* heads: each IP layer's bottom diff is computed separately, the auto-
   * inserted Split layer adds them. */
  float *tmp = (float *)malloc((size_t)n * H * sizeof(float));
  int col = 0;
  for (int h = 0; h < l->n_heads; ++h) {
    float *dst = (h == 0) ? dacts[L] : tmp;
    ip_dgrad(n, l->head_out[h], H, d_head + col, ld_head, w + l->w_off[L + h], dst, H);
    if (h > 0)
      for (size_t i = 0; i < (size_t)n * H; ++i) dacts[L][i] = dacts[L][i] + tmp[i];
    if (want_wgrad)
      ip_wgrad(n, l->head_out[h], H, d_head + col, ld_head, acts[L], H,
               g + l->w_off[L + h], g + l->b_off[L + h]);
    col += l->head_out[h];
  }
  free(tmp);
  for (int i = L - 1; i >= 0; --i) {
    const int N = l->dims[i + 1], K = l->dims[i];
    /* ReLU backward (in place on the diff) */
    float *dz = dacts[i + 1];
    const float *yv = acts[i + 1];
    for (size_t e = 0; e < (size_t)n * N; ++e) dz[e] = lrelu_bwd(dz[e], yv[e]);
    if (want_wgrad)
      ip_wgrad(n, N, K, dz, N, acts[i], K, g + l->w_off[i], g + l->b_off[i]);
    if (i > 0 || want_input_grad)
      ip_dgrad(n, N, K, dz, N, w + l->w_off[i], dacts[i], K);
  }
}

/* SGDSolver::ClipGradients + AdamSolver::ComputeUpdateValue + Net::Update
 * (Caffe sgd_solver.cpp / adam_solver.cpp @2ef5847).
 * Reached from critic_solver_->Step(1) (src/dqn.cpp:904) and
 * actor_solver_->ApplyUpdate() (src/dqn.cpp:964).  iter is the value BEFORE
 * the increment (t = iter + 1). */
static void solver_apply(const orc_layout *l, float *w, float *g, float *m, float *v, int iter,
                         float lr, const orc_config *c) {
  const size_t P = l->count;
  if (c->clip >= 0.0f) {
    /* ClipGradients: sumsq_diff += net_params[i]->sumsq_diff() — one cblas_sdot per
     * param blob, accumulated over blobs in Dtype.  The summation order inside sdot
     * is BLAS-implementation-defined, so each blob's sum is taken in double (the
     * limit every reasonable order converges to) and rounded to float once. */
    float sumsq = 0.0f;
    const int nblob = l->L + l->n_heads;
    for (int i = 0; i < nblob; ++i) {
      const int rows = i < l->L ? l->dims[i + 1] : l->head_out[i - l->L];
      const int cols = i < l->L ? l->dims[i] : l->dims[l->L];
      double sw = 0.0, sb = 0.0;
      const float *gw = g + l->w_off[i], *gb = g + l->b_off[i];
      for (size_t e = 0; e < (size_t)rows * cols; ++e) sw += (double)gw[e] * (double)gw[e];
      for (int e = 0; e < rows; ++e) sb += (double)gb[e] * (double)gb[e];
      sumsq += (float)sw;
      sumsq += (float)sb;
    }
    const float l2 = sqrtf(sumsq);
    if (l2 > c->clip) {
      const float s = c->clip / l2;
      for (size_t i = 0; i < P; ++i) g[i] *= s;
    }
  }
  const int t = iter + 1;
  const float b1 = c->beta1, b2 = c->beta2;
  /* std::sqrt(Dtype(1) - pow(beta2, t)) / (Dtype(1.) - pow(beta1, t)): pow(float,int)
   * promotes to double in C++11, so the expression is evaluated in double and
   * rounded once to Dtype. */
  const float correction =
      (float)(sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  const float step = lr * correction;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < P; ++i) {
    const float gi = g[i];
    /* caffe_cpu_axpby(N, 1-b1, g, b1, m): m = b1*m, then m += (1-b1)*g (axpy as FMA) */
    float mi = fmaf(omb1, gi, b1 * m[i]);
    float vi = fmaf(omb2, gi * gi, b2 * v[i]);
    m[i] = mi; v[i] = vi;
    float upd = step * (mi / (sqrtf(vi) + c->eps));
    g[i] = upd;             /* Caffe leaves the update value in diff */
    w[i] = w[i] - upd;      /* Net::Update: data -= diff */
  }
}

/* DQN::SoftUpdateNet (src/dqn.cpp:1085-1096): caffe_cpu_axpby(N, tau, from,
 * 1-tau, to) with tau a float: to = (1-tau)*to, then to += tau*from. */
static void soft_update(size_t P, const float *from, float *to, float tau) {
  const float omt = 1 - tau;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < P; ++i) to[i] = fmaf(tau, from[i], omt * to[i]);
}

/* ------------------------------------------------------------------ object */

static float *zalloc(size_t n) { return (float *)calloc(n ? n : 1, sizeof(float)); }

orc *orc_create(const orc_config *cfg) {
  orc *o = (orc *)calloc(1, sizeof(orc));
  o->cfg = *cfg;
  if (o->cfg.global_B <= 0) o->cfg.global_B = cfg->B;
  layout_init(&o->la, cfg->S, cfg, 1);
  layout_init(&o->lc, cfg->S + ORC_NO, cfg, 0);
  for (int i = 0; i < 4; ++i) o->w[i] = zalloc((i & 1) ? o->lc.count : o->la.count);
  for (int i = 0; i < 2; ++i) {
    size_t P = i ? o->lc.count : o->la.count;
    o->g[i] = zalloc(P); o->m[i] = zalloc(P); o->v[i] = zalloc(P);
  }
  const size_t cap = cfg->capacity, S = cfg->S, B = cfg->B;
  o->r_state = zalloc(cap * S); o->r_next = zalloc(cap * S);
  o->r_act = zalloc(cap * ORC_NO); o->r_rew = zalloc(cap); o->r_mc = zalloc(cap);
  o->r_term = (uint8_t *)calloc(cap ? cap : 1, 1);
  o->mb_s = zalloc(B * S); o->mb_n = zalloc(B * S); o->mb_a = zalloc(B * ORC_NO);
  o->mb_r = zalloc(B); o->mb_mc = zalloc(B); o->mb_t = (uint8_t *)calloc(B, 1);
  o->mb_idx = (int32_t *)calloc(B, sizeof(int32_t));
  o->q_target = zalloc(B); o->y = zalloc(B); o->q_train = zalloc(B);
  o->q_policy = zalloc(B); o->actor_out = zalloc(B * ORC_NO); o->dq_da = zalloc(B * ORC_NO);
  for (int l = 0; l <= cfg->L; ++l) {
    o->actA[l] = zalloc(B * o->la.dims[l]); o->dA[l] = zalloc(B * o->la.dims[l]);
    o->actC[l] = zalloc(B * o->lc.dims[l]); o->dC[l] = zalloc(B * o->lc.dims[l]);
  }
  return o;
}

void orc_destroy(orc *o) {
  if (!o) return;
  for (int i = 0; i < 4; ++i) free(o->w[i]);
  for (int i = 0; i < 2; ++i) { free(o->g[i]); free(o->m[i]); free(o->v[i]); }
  free(o->r_state); free(o->r_next); free(o->r_act); free(o->r_rew); free(o->r_mc);
  free(o->r_term); free(o->mb_s); free(o->mb_n); free(o->mb_a); free(o->mb_r);
  free(o->mb_mc); free(o->mb_t); free(o->mb_idx); free(o->q_target); free(o->y);
  free(o->q_train); free(o->q_policy); free(o->actor_out); free(o->dq_da);
  for (int l = 0; l <= o->cfg.L; ++l) {
    free(o->actA[l]); free(o->dA[l]); free(o->actC[l]); free(o->dC[l]);
  }
  free(o);
}

size_t orc_param_count(const orc *o, int net) { return (net & 1) ? o->lc.count : o->la.count; }

/* kind: 0 w, 1 m, 2 v, 3 g */
static float *param_ptr(orc *o, int net, int kind) {
  if (kind == 0) return (net >= 0 && net < 4) ? o->w[net] : NULL;
  if (net < 0 || net > 1) return NULL;
  return kind == 1 ? o->m[net] : kind == 2 ? o->v[net] : kind == 3 ? o->g[net] : NULL;
}
int orc_get_params(orc *o, int net, int kind, float *dst) {
  float *p = param_ptr(o, net, kind); if (!p) return 1;
  memcpy(dst, p, orc_param_count(o, net) * sizeof(float)); return 0;
}
int orc_set_params(orc *o, int net, int kind, const float *src) {
  float *p = param_ptr(o, net, kind); if (!p) return 1;
  memcpy(p, src, orc_param_count(o, net) * sizeof(float)); return 0;
}
/* CloneNet (src/dqn.cpp:1022-1035): hard copy online -> target */
void orc_clone_to_target(orc *o, int net) {
  memcpy(o->w[net + 2], o->w[net], orc_param_count(o, net) * sizeof(float));
}
void orc_get_iters(const orc *o, int *a, int *c) { *a = o->iter[0]; *c = o->iter[1]; }
void orc_set_iters(orc *o, int a, int c) { o->iter[0] = a; o->iter[1] = c; }
float *orc_grad_ptr(orc *o, int net) { return o->g[net]; }
float *orc_tail_ptr(orc *o, int net) { return o->tail[net]; }

/* ------------------------------------------------------------ replay memory */

static inline int64_t phys(const orc *o, int64_t logical) {
  return (o->head + logical) % o->cfg.capacity;
}
static void ring_pop_front(orc *o) { o->head = (o->head + 1) % o->cfg.capacity; o->size--; }
static void ring_push_back(orc *o, const float *s, const float *a, float r, float mc,
                           const float *nx, uint8_t term) {
  const int S = o->cfg.S;
  const int64_t p = phys(o, o->size);
  memcpy(o->r_state + p * S, s, S * sizeof(float));
  memcpy(o->r_act + p * ORC_NO, a, ORC_NO * sizeof(float));
  o->r_rew[p] = r; o->r_mc[p] = mc; o->r_term[p] = term ? 1 : 0;
  if (!term && nx) memcpy(o->r_next + p * S, nx, S * sizeof(float));
  else memset(o->r_next + p * S, 0, S * sizeof(float));
  o->size++;
}

/* DQN::AddTransition (src/dqn.cpp:768-773) */
void orc_add_transition(orc *o, const float *s, const float *a, float r, float mc,
                        const float *nx, uint8_t term) {
  if (o->size == o->cfg.capacity) ring_pop_front(o);
  ring_push_back(o, s, a, r, mc, nx, term);
}

/* DQN::AddTransitions (src/dqn.cpp:775-781): pops while size + n >= capacity
 * (so at most capacity-1 remain), then inserts all n at the end. */
int orc_add_transitions(orc *o, const float *s, const float *a, const float *r,
                        const float *mc, const float *nx, const uint8_t *term, int n) {
  const int S = o->cfg.S;
  while (o->size + n >= o->cfg.capacity) {
    if (o->size == 0) return 1; /* the reference would pop an empty deque (UB) */
    ring_pop_front(o);
  }
  for (int i = 0; i < n; ++i)
    ring_push_back(o, s + (size_t)i * S, a + (size_t)i * ORC_NO, r[i], mc[i],
                   nx ? nx + (size_t)i * S : NULL, term[i]);
  return 0;
}
int orc_memory_size(const orc *o) { return (int)o->size; }
void orc_clear_memory(orc *o) { o->head = 0; o->size = 0; }
void orc_read_memory(const orc *o, int first, int n, float *s, float *a, float *r,
                     float *mc, float *nx, uint8_t *term) {
  const int S = o->cfg.S;
  for (int i = 0; i < n; ++i) {
    const int64_t p = phys(o, first + i);
    if (s) memcpy(s + (size_t)i * S, o->r_state + p * S, S * sizeof(float));
    if (nx) memcpy(nx + (size_t)i * S, o->r_next + p * S, S * sizeof(float));
    if (a) memcpy(a + (size_t)i * ORC_NO, o->r_act + p * ORC_NO, ORC_NO * sizeof(float));
    if (r) r[i] = o->r_rew[p];
    if (mc) mc[i] = o->r_mc[p];
    if (term) term[i] = o->r_term[p];
  }
}

/* DQN::LabelTransitions (src/dqn.cpp:783-797): reverse scan, gamma_ is a
 * double (src/dqn.hpp:186) so r + gamma*target is evaluated in double and
 * rounded to float on store. */
void orc_label_transitions(double gamma, const float *rewards, int n, float *mc) {
  if (n <= 0) return;
  mc[n - 1] = rewards[n - 1];
  for (int i = n - 2; i >= 0; --i) mc[i] = (float)((double)rewards[i] + gamma * (double)mc[i + 1]);
}

/* --------------------------------------------------------- acting-time path */

/* DQN::SelectActionGreedily (src/dqn.cpp:734-766): actor forward.  The
 * reference zero-pads to kMinibatchSize rows; rows are independent, so only
 * the n real rows are computed. */
void orc_actor_forward(orc *o, int net, const float *states, int n, float *out) {
  const orc_layout *l = &o->la;
  float *acts[ORC_MAXL + 1];
  acts[0] = (float *)states;
  for (int i = 1; i <= l->L; ++i) acts[i] = (float *)malloc((size_t)n * l->dims[i] * sizeof(float));
  net_forward(l, o->w[net], acts, n, out, ORC_NO);
  for (int i = 1; i <= l->L; ++i) free(acts[i]);
}

/* DQN::CriticForward (src/dqn.cpp:982-1020): Concat(states, actions,
 * action_params) on axis 2 (src/dqn.cpp:446-448) then tower then q_values. */
void orc_critic_forward(orc *o, int net, const float *states, const float *actor_out,
                        int n, float *q) {
  const orc_layout *l = &o->lc;
  const int S = o->cfg.S;
  float *acts[ORC_MAXL + 1];
  acts[0] = (float *)malloc((size_t)n * l->in_dim * sizeof(float));
  for (int i = 0; i < n; ++i) {
    memcpy(acts[0] + (size_t)i * l->in_dim, states + (size_t)i * S, S * sizeof(float));
    memcpy(acts[0] + (size_t)i * l->in_dim + S, actor_out + (size_t)i * ORC_NO, ORC_NO * sizeof(float));
  }
  for (int i = 1; i <= l->L; ++i) acts[i] = (float *)malloc((size_t)n * l->dims[i] * sizeof(float));
  net_forward(l, o->w[net], acts, n, q, 1);
  for (int i = 0; i <= l->L; ++i) free(acts[i]);
}

/* GetParamOffset (src/dqn.cpp:162-178) */
static int param_offset(int action, int arg_num) {
  if (arg_num < 0 || arg_num > 1) return -1;
  switch (action) {
    case 0: return arg_num;                 /* DASH   */
    case 1: return arg_num == 0 ? 2 : -1;   /* TURN   */
    case 2: return arg_num == 0 ? 3 : -1;   /* TACKLE */
    case 3: return 4 + arg_num;             /* KICK   */
  }
  return -1;
}
/* GetAction (src/dqn.cpp:196-208): TACKLE masked to -99999, argmax over the 4
 * logits (std::max_element: first maximum wins), pick the action's params. */
void orc_get_action(const float *actor_out, int n, int32_t *action, float *arg1, float *arg2) {
  for (int i = 0; i < n; ++i) {
    float c[ORC_NA];
    for (int j = 0; j < ORC_NA; ++j) c[j] = actor_out[(size_t)i * ORC_NO + j];
    c[2] = -99999.0f;
    int best = 0;
    for (int j = 1; j < ORC_NA; ++j) if (c[j] > c[best]) best = j;
    action[i] = best;
    arg1[i] = actor_out[(size_t)i * ORC_NO + ORC_NA + param_offset(best, 0)];
    const int o2 = param_offset(best, 1);
    arg2[i] = o2 < 0 ? 0.0f : actor_out[(size_t)i * ORC_NO + ORC_NA + o2];
  }
}

/* ------------------------------------------------------------- the hot path */

/* inverting gradients (src/dqn.cpp:927-957) on one element */
static inline float invert_grad(float diff, float output, float mn, float mx) {
  if (diff < 0) diff *= (mx - output) / (mx - mn);
  else if (diff > 0) diff *= (output - mn) / (mx - mn);
  return diff;
}

/* DQN::UpdateActorCritic (src/dqn.cpp:828-972), cut into the three phases the
 * data-parallel form needs (phase boundaries are where gradients are
 * complete).  idx: B logical replay indices (the explicit form of
 * SampleTransitionsFromMemory, src/dqn.cpp:501-509). */
int orc_update_phase(orc *o, int phase, const int32_t *idx) {
  const orc_config *c = &o->cfg;
  const int B = c->B, S = c->S;
  const orc_layout *la = &o->la, *lc = &o->lc;
  const int Kc = lc->in_dim;
  if (phase == 0) {
    /* gather (src/dqn.cpp:859-887) */
    for (int n = 0; n < B; ++n) {
      if (idx[n] < 0 || idx[n] >= o->size) return 2;
      const int64_t p = phys(o, idx[n]);
      o->mb_idx[n] = idx[n];
      memcpy(o->mb_s + (size_t)n * S, o->r_state + p * S, S * sizeof(float));
      memcpy(o->mb_a + (size_t)n * ORC_NO, o->r_act + p * ORC_NO, ORC_NO * sizeof(float));
      o->mb_r[n] = o->r_rew[p]; o->mb_mc[n] = o->r_mc[p]; o->mb_t[n] = o->r_term[p];
      memcpy(o->mb_n + (size_t)n * S, o->r_next + p * S, S * sizeof(float));
    }
    /* target_q = CriticForwardThroughActor(critic_target, actor_target, next)
     * (src/dqn.cpp:889-891, 974-980).  The reference compacts non-terminal
     * rows; rows are independent so all B rows are evaluated and terminal rows
     * ignored (or skipped entirely unless mirror_waste). */
    float *mu_next = zalloc((size_t)B * ORC_NO);
    orc_actor_forward(o, 2, o->mb_n, B, mu_next);
    orc_critic_forward(o, 3, o->mb_n, mu_next, B, o->q_target);
    free(mu_next);
    /* TD target (src/dqn.cpp:892-900): doubles exactly where the reference has them */
    for (int n = 0; n < B; ++n) {
      const float off_policy = o->mb_t[n]
          ? o->mb_r[n]
          : (float)((double)o->mb_r[n] + c->gamma * (double)o->q_target[n]);
      const float target = (float)(c->beta * (double)o->mb_mc[n] + (1 - c->beta) * (double)off_policy);
      if (!isfinite(target)) return 3;   /* CHECK(std::isfinite(target)) :898 */
      o->y[n] = target;
    }
    /* critic_solver_->Step(1) (src/dqn.cpp:904): ClearParamDiffs, forward+loss,
     * backward */
    memset(o->g[1], 0, lc->count * sizeof(float));
    for (int n = 0; n < B; ++n) {
      memcpy(o->actC[0] + (size_t)n * Kc, o->mb_s + (size_t)n * S, S * sizeof(float));
      memcpy(o->actC[0] + (size_t)n * Kc + S, o->mb_a + (size_t)n * ORC_NO, ORC_NO * sizeof(float));
    }
    net_forward(lc, o->w[1], o->actC, B, o->q_train, 1);
    /* EuclideanLoss: loss = sum(d^2)/num/2 ; bottom diff = d/num.
     * num = global batch under data parallelism. */
    float dot = 0.0f;
    float *dq = zalloc(B);
    const float alpha = 1.0f / (float)c->global_B;
    for (int n = 0; n < B; ++n) {
      const float d = o->q_train[n] - o->y[n];
      dot = fmaf(d, d, dot);
      dq[n] = alpha * d;
    }
    o->tail[1][0] = dot / (float)c->global_B / 2.0f;
    net_backward(lc, o->w[1], o->g[1], o->actC, o->dC, B, dq, 1, 1, c->mirror_waste);
    free(dq);
    return 0;
  }
  if (phase == 1) {
    /* rest of Step(1): clip, Adam, update, ++iter */
    solver_apply(lc, o->w[1], o->g[1], o->m[1], o->v[1], o->iter[1], c->lr_critic, c);
    o->iter[1] += 1;
    o->last_loss = o->tail[1][0];
    if (!isfinite(o->last_loss)) return 4; /* CHECK(isfinite(critic_loss)) :906 */
    /* ZeroGradParameters x2 (src/dqn.cpp:908-909) */
    memset(o->g[1], 0, lc->count * sizeof(float));
    memset(o->g[0], 0, la->count * sizeof(float));
    /* actor forward on the sampled states (src/dqn.cpp:910-911) */
    memcpy(o->actA[0], o->mb_s, (size_t)B * S * sizeof(float));
    net_forward(la, o->w[0], o->actA, B, o->actor_out, ORC_NO);
    /* critic forward on (s, mu(s)) with the UPDATED critic (src/dqn.cpp:913-916) */
    for (int n = 0; n < B; ++n) {
      memcpy(o->actC[0] + (size_t)n * Kc, o->mb_s + (size_t)n * S, S * sizeof(float));
      memcpy(o->actC[0] + (size_t)n * Kc + S, o->actor_out + (size_t)n * ORC_NO, ORC_NO * sizeof(float));
    }
    net_forward(lc, o->w[1], o->actC, B, o->q_policy, 1);
    double qs = 0.0;
    for (int n = 0; n < B; ++n) qs += (double)o->q_policy[n];   /* std::accumulate(.., 0.0) */
    o->tail[0][1] = (float)qs;
    o->last_avgq = (float)(qs / (float)c->global_B);
    /* q_values diff = -1 per row; critic BackwardFrom(q_values_layer)
     * (src/dqn.cpp:918-923).  The reference also computes (and later discards)
     * every critic dW here. */
    float *dq = zalloc(B);
    for (int n = 0; n < B; ++n) dq[n] = -1.0f;
    net_backward(lc, o->w[1], o->g[1], o->actC, o->dC, B, dq, 1, c->mirror_waste, 1);
    free(dq);
    /* inverting gradients (src/dqn.cpp:924-957) */
    for (int n = 0; n < B; ++n) {
      for (int h = 0; h < ORC_NA; ++h) {
        const float diff = o->dC[0][(size_t)n * Kc + S + h];
        o->dq_da[(size_t)n * ORC_NO + h] =
            invert_grad(diff, o->actor_out[(size_t)n * ORC_NO + h], -1.0f, 1.0f);
      }
      for (int h = 0; h < ORC_NP; ++h) {
        const float diff = o->dC[0][(size_t)n * Kc + S + ORC_NA + h];
        float mn, mx;
        if (h == 0 || h == 4) { mn = 0; mx = 100; } else { mn = -180; mx = 180; }
        o->dq_da[(size_t)n * ORC_NO + ORC_NA + h] =
            invert_grad(diff, o->actor_out[(size_t)n * ORC_NO + ORC_NA + h], mn, mx);
      }
    }
    /* ShareDiff + actor BackwardFrom("actionpara_layer") (src/dqn.cpp:960-963) */
    net_backward(la, o->w[0], o->g[0], o->actA, o->dA, B, o->dq_da, ORC_NO, 1, c->mirror_waste);
    return 0;
  }
  if (phase == 2) {
    /* actor_solver_->ApplyUpdate(); set_iter(iter+1) (src/dqn.cpp:964-965) */
    solver_apply(la, o->w[0], o->g[0], o->m[0], o->v[0], o->iter[0], c->lr_actor, c);
    o->iter[0] += 1;
    /* soft target update (src/dqn.cpp:967-970) */
    const int mx = o->iter[0] > o->iter[1] ? o->iter[0] : o->iter[1];
    if (mx % c->soft_update_freq == 0) {
      soft_update(lc->count, o->w[1], o->w[3], c->tau);
      soft_update(la->count, o->w[0], o->w[2], c->tau);
    }
    return 0;
  }
  return 1;
}

/* One solver's ApplyUpdate() in isolation, on the gradient currently in g[net]: ClipGradients +
 * Adam + Net::Update (actor_solver_->ApplyUpdate(), src/dqn.cpp:964; the tail of
 * critic_solver_->Step(1), :904), the soft update of THAT net's target if max_iter() after both
 * increments of a full update would be a multiple of soft_update_freq (:967-970), then ++iter of
 * that net (:965).  Lets the optimiser pass be compared on identical (w, g, m, v, w', iter). */
int orc_apply_update(orc *o, int net) {
  const orc_config *c = &o->cfg;
  if (net != 0 && net != 1) return 1;
  const orc_layout *l = net ? &o->lc : &o->la;
  const int mx = (o->iter[0] + 1) > (o->iter[1] + 1) ? (o->iter[0] + 1) : (o->iter[1] + 1);
  solver_apply(l, o->w[net], o->g[net], o->m[net], o->v[net], o->iter[net], net ? c->lr_critic : c->lr_actor, c);
  if (mx % c->soft_update_freq == 0) soft_update(l->count, o->w[net], o->w[net + 2], c->tau);
  o->iter[net] += 1;
  return 0;
}

int orc_update(orc *o, const int32_t *idx, float *loss, float *avgq) {
  for (int p = 0; p < 3; ++p) { int rc = orc_update_phase(o, p, idx); if (rc) return rc; }
  if (loss) *loss = o->last_loss;
  if (avgq) *avgq = o->last_avgq;
  return 0;
}
/* after an external all-reduce of the tails: recompute the reported scalars */
void orc_set_stats_from_tails(orc *o) {
  o->last_loss = o->tail[1][0];
  o->last_avgq = (float)((double)o->tail[0][1] / (float)o->cfg.global_B);
}
void orc_last_stats(const orc *o, float *loss, float *avgq) { *loss = o->last_loss; *avgq = o->last_avgq; }

int orc_debug_read(orc *o, const char *name, float *dst, size_t count) {
  const size_t B = o->cfg.B;
  const float *src = NULL; size_t n = 0;
  if (!strcmp(name, "q_target")) { src = o->q_target; n = B; }
  else if (!strcmp(name, "y")) { src = o->y; n = B; }
  else if (!strcmp(name, "q_train")) { src = o->q_train; n = B; }
  else if (!strcmp(name, "q_policy")) { src = o->q_policy; n = B; }
  else if (!strcmp(name, "actor_out")) { src = o->actor_out; n = B * ORC_NO; }
  else if (!strcmp(name, "dq_da")) { src = o->dq_da; n = B * ORC_NO; }
  else if (!strcmp(name, "idx")) { if (count < B) return 2; for (size_t i = 0; i < B; ++i) dst[i] = (float)o->mb_idx[i]; return 0; }
  else if (!strcmp(name, "terminal")) { if (count < B) return 2; for (size_t i = 0; i < B; ++i) dst[i] = (float)o->mb_t[i]; return 0; }
  else if ((!strncmp(name, "actA_", 5) || !strncmp(name, "actC_", 5))) {
    /* stored (post-ReLU) tower activations of the LAST actor(s) / critic forward of the update in progress: actC holds the
     * training forward critic(s, a) after phase 0 and critic(s, mu(s)) after phase 1.  Parity tests compare signs. */
    const int critic = name[3] == 'C', i = atoi(name + 5);
    const orc_layout *l = critic ? &o->lc : &o->la;
    if (i < 1 || i > l->L) return 1;
    src = (critic ? o->actC : o->actA)[i]; n = B * (size_t)l->dims[i];
  }
  else return 1;
  if (count < n) return 2;
  memcpy(dst, src, n * sizeof(float));
  return 0;
}

/* --------------------------------------------- HFOGameState reward shaping */

/* Carried per-worker state of HFOGameState (src/hfo_game.hpp:29-60). */
typedef struct {
  float old_ball_prox, ball_prox_delta, old_kickable, kickable_delta,
        old_ball_dist_goal, ball_dist_goal_delta;
  int32_t steps, episode_over, got_kickable_reward, pass_active;
  int32_t player_on_ball_unum, old_player_on_ball_unum, our_unum, status;
  double total_reward, extrinsic_reward;
} orc_game;

/* HFOGameState::update (src/hfo_game.cpp:122-173) minus the hfo.step() I/O:
 * `status` and `player_on_ball` are inputs (what the HFO server would return).
 * status: 0 IN_GAME, 1 GOAL, 2 CAPTURED_BY_DEFENSE, 3 OUT_OF_BOUNDS, 4 OUT_OF_TIME. */
void orc_game_update(orc_game *g, const float *state, int status, int player_on_ball) {
  g->status = status;
  if (status != 0) g->episode_over = 1;
  const float ball_proximity = state[53], goal_proximity = state[15];
  const float ball_dist = 1.0 - ball_proximity, goal_dist = 1.0 - goal_proximity;
  const float kickable = state[12];
  float ball_ang_rad = acos(state[52]);
  if (state[51] < 0) ball_ang_rad *= -1.;
  float goal_ang_rad = acos(state[14]);
  if (state[13] < 0) goal_ang_rad *= -1.;
  const float alpha = fmaxf(ball_ang_rad, goal_ang_rad) - fminf(ball_ang_rad, goal_ang_rad);
  const float ball_dist_goal =
      sqrt(ball_dist * ball_dist + goal_dist * goal_dist - 2. * ball_dist * goal_dist * cos(alpha));
  const float ball_vel_valid = state[54], ball_vel = state[55];
  if (ball_vel_valid && ball_vel > -.5 /* kPassVelThreshold, src/hfo_game.hpp:18 */) g->pass_active = 1;
  if (g->steps > 0) {
    g->ball_prox_delta = ball_proximity - g->old_ball_prox;
    g->kickable_delta = kickable - g->old_kickable;
    g->ball_dist_goal_delta = ball_dist_goal - g->old_ball_dist_goal;
  }
  g->old_ball_prox = ball_proximity; g->old_kickable = kickable;
  g->old_ball_dist_goal = ball_dist_goal;
  if (g->episode_over) { g->ball_prox_delta = 0; g->kickable_delta = 0; g->ball_dist_goal_delta = 0; }
  g->old_player_on_ball_unum = g->player_on_ball_unum;
  g->player_on_ball_unum = player_on_ball;
  g->steps++;
}

/* HFOGameState::reward (src/hfo_game.cpp:175-236): moveToBall + 3*kickToGoal +
 * EOT; pass_reward() is evaluated (it mutates pass_active) but not added
 * (:178-180). */
float orc_game_reward(orc_game *g) {
  float mtb = 0;
  if (g->player_on_ball_unum < 0 || g->player_on_ball_unum == g->our_unum) mtb += g->ball_prox_delta;
  if (g->kickable_delta >= 1 && !g->got_kickable_reward) { mtb += 1.0; g->got_kickable_reward = 1; }
  float ktg = 0;
  if (g->player_on_ball_unum == g->our_unum) ktg = -g->ball_dist_goal_delta;
  else if (g->got_kickable_reward) ktg = 0.2 * -g->ball_dist_goal_delta;
  const float kickToGoal = 3. * ktg;
  if (g->pass_active && g->player_on_ball_unum > 0 &&
      g->player_on_ball_unum != g->old_player_on_ball_unum) g->pass_active = 0;
  float eot = 0;
  if (g->status == 1) eot = (g->player_on_ball_unum == g->our_unum) ? 5 : 1;
  const float reward = mtb + kickToGoal + eot;
  g->extrinsic_reward += eot; g->total_reward += reward;
  return reward;
}

/* ------------------------------------------------- batched env front-end (oracle side) */
/* CPU restatement of PlayOneEpisode's learner side (src/dqn_main.cpp:97-153) for N workers on
 * the SAME synthetic state stream as dqn-hfo_amd/csrc/env.hip.h (counter-based Philox-4x32-10
 * keyed by (seed, per-worker draw counter, worker*256 + draw index)).  The synthetic stream
 * stands in for rcssserver; everything after it is the reference's logic. */

static void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
static uint32_t philox_u32(uint64_t seed, uint64_t ctr, uint32_t lane) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), lane, 0x9E3779B9u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  return c[0];
}
/* the learner's on-device sampler (SampleTransitionsFromMemory stand-in), for tests */
int32_t orc_philox_index(uint64_t seed, uint64_t update_counter, uint32_t row, int32_t size) {
  return (int32_t)(((uint64_t)philox_u32(seed, update_counter, row) * (uint64_t)size) >> 32);
}

typedef struct {
  orc *o;
  int N, T, unum;
  float p_end, p_goal;
  uint64_t seed;
  float *cur;            /* [N][S] */
  float *ep_s, *ep_a, *ep_r;
  orc_game *game;
  int *len;
  uint64_t *g;
  int32_t *act; float *arg1, *arg2, *rew;
  int64_t n_steps, n_episodes, n_goals; double reward_sum;
} orc_env;

static float env_u01(const orc_env *e, uint64_t g, int w, int k) {
  return (float)(philox_u32(e->seed, g, (uint32_t)(w * 256 + k)) >> 8) * (1.0f / 16777216.0f);
}
static float env_feature(const orc_env *e, uint64_t g, int w, int f) {
  const float kPi = 3.14159265358979323846f;
  if (f == 13 || f == 14) { const float th = fmaf(2.0f, env_u01(e, g, w, 16 + 13), -1.0f) * kPi; return f == 13 ? sinf(th) : cosf(th); }
  if (f == 51 || f == 52) { const float th = fmaf(2.0f, env_u01(e, g, w, 16 + 51), -1.0f) * kPi; return f == 51 ? sinf(th) : cosf(th); }
  const float u = env_u01(e, g, w, 16 + f);
  if (f == 12 || f == 54) return u < 0.5f ? -1.0f : 1.0f;
  return fmaf(2.0f, u, -1.0f);
}
static void env_reset_worker(orc_env *e, int w) {
  const int S = e->o->cfg.S;
  for (int f = 0; f < S; ++f) e->cur[(size_t)w * S + f] = env_feature(e, e->g[w], w, f);
  memset(&e->game[w], 0, sizeof(orc_game));
  e->game[w].our_unum = e->unum;
  orc_game_update(&e->game[w], e->cur + (size_t)w * S, 0, 0);   /* after the forced DASH(0,0), src/dqn_main.cpp:103-105 */
  e->len[w] = 0; e->g[w] += 1;
}

orc_env *orc_env_create(orc *o, int workers, int max_steps, int unum, float p_end, float p_goal, uint64_t seed) {
  orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
  const size_t N = workers, S = o->cfg.S, T = max_steps;
  e->o = o; e->N = workers; e->T = max_steps; e->unum = unum; e->p_end = p_end; e->p_goal = p_goal; e->seed = seed;
  e->cur = zalloc(N * S); e->ep_s = zalloc(N * T * S); e->ep_a = zalloc(N * T * ORC_NO); e->ep_r = zalloc(N * T);
  e->game = (orc_game *)calloc(N, sizeof(orc_game)); e->len = (int *)calloc(N, sizeof(int));
  e->g = (uint64_t *)calloc(N, sizeof(uint64_t));
  e->act = (int32_t *)calloc(N, sizeof(int32_t)); e->arg1 = zalloc(N); e->arg2 = zalloc(N); e->rew = zalloc(N);
  for (int w = 0; w < workers; ++w) env_reset_worker(e, w);
  return e;
}
void orc_env_destroy(orc_env *e) {
  if (!e) return;
  free(e->cur); free(e->ep_s); free(e->ep_a); free(e->ep_r); free(e->game); free(e->len); free(e->g);
  free(e->act); free(e->arg1); free(e->arg2); free(e->rew); free(e);
}

void orc_env_step(orc_env *e, float epsilon, int n_steps) {
  orc *o = e->o;
  const int S = o->cfg.S, N = e->N, T = e->T;
  float *greedy = zalloc((size_t)N * ORC_NO);
  float *next = zalloc(S);
  for (int s = 0; s < n_steps; ++s) {
    orc_actor_forward(o, 0, e->cur, N, greedy);            /* SelectActionGreedily, batched */
    int *done = (int *)calloc(N, sizeof(int));
    for (int w = 0; w < N; ++w) {
      const uint64_t g = e->g[w];
      const int len = e->len[w];
      float ao[ORC_NO];
      if (env_u01(e, g, w, 0) < epsilon) {                /* one epsilon draw per SelectAction call */
        for (int j = 0; j < ORC_NO; ++j) {                /* GetRandomActorOutput ranges */
          const float u = env_u01(e, g, w, 1 + j);
          if (j < ORC_NA) ao[j] = fmaf(2.0f, u, -1.0f);
          else if (j == ORC_NA + 0) ao[j] = fmaf(200.0f, u, -100.0f);
          else if (j == ORC_NA + 4) ao[j] = 100.0f * u;
          else ao[j] = fmaf(360.0f, u, -180.0f);
        }
      } else memcpy(ao, greedy + (size_t)w * ORC_NO, sizeof ao);
      orc_get_action(ao, 1, &e->act[w], &e->arg1[w], &e->arg2[w]);
      memcpy(e->ep_s + ((size_t)w * T + len) * S, e->cur + (size_t)w * S, S * sizeof(float));
      memcpy(e->ep_a + ((size_t)w * T + len) * ORC_NO, ao, sizeof ao);
      for (int f = 0; f < S; ++f) next[f] = env_feature(e, g, w, f);
      int status = 0;
      if (env_u01(e, g, w, 11) < e->p_end) status = env_u01(e, g, w, 12) < e->p_goal ? 1 : 2;
      if (status == 0 && len + 1 >= T) status = 4;
      const int pob = env_u01(e, g, w, 13) < 0.5f ? e->unum : -1;
      orc_game_update(&e->game[w], next, status, pob);
      const float r = orc_game_reward(&e->game[w]);
      e->ep_r[(size_t)w * T + len] = r; e->rew[w] = r;
      memcpy(e->cur + (size_t)w * S, next, S * sizeof(float));
      e->len[w] = len + 1; e->g[w] = g + 1;
      e->n_steps += 1; e->reward_sum += (double)r;
      if (status == 1) e->n_goals += 1;
      done[w] = status != 0;
    }
    for (int w = 0; w < N; ++w) {                         /* LabelTransitions + AddTransitions, worker order */
      if (!done[w]) continue;
      const int len = e->len[w];
      float *mc = zalloc(len), *nx = zalloc((size_t)len * S);
      uint8_t *term = (uint8_t *)calloc(len, 1);
      orc_label_transitions(o->cfg.gamma, e->ep_r + (size_t)w * T, len, mc);
      for (int t = 0; t + 1 < len; ++t) memcpy(nx + (size_t)t * S, e->ep_s + ((size_t)w * T + t + 1) * S, S * sizeof(float));
      term[len - 1] = 1;
      orc_add_transitions(o, e->ep_s + (size_t)w * T * S, e->ep_a + (size_t)w * T * ORC_NO, e->ep_r + (size_t)w * T, mc, nx, term, len);
      free(mc); free(nx); free(term);
      e->n_episodes += 1;
      env_reset_worker(e, w);
    }
    free(done);
  }
  free(greedy); free(next);
}
void orc_env_stats(const orc_env *e, int64_t *steps, int64_t *episodes, double *reward_sum, int64_t *goals) {
  *steps = e->n_steps; *episodes = e->n_episodes; *reward_sum = e->reward_sum; *goals = e->n_goals;
}
void orc_env_read(const orc_env *e, int32_t *act, float *arg1, float *arg2, float *rew, float *state, int32_t *len) {
  const size_t N = e->N, S = e->o->cfg.S;
  if (act) memcpy(act, e->act, N * 4);
  if (arg1) memcpy(arg1, e->arg1, N * 4);
  if (arg2) memcpy(arg2, e->arg2, N * 4);
  if (rew) memcpy(rew, e->rew, N * 4);
  if (state) memcpy(state, e->cur, N * S * 4);
  if (len) for (size_t i = 0; i < N; ++i) len[i] = e->len[i];
}
