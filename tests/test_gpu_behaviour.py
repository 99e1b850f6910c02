"""Behaviour of the learner that does NOT pass through `oracle/`.

Parity against the reference itself cannot be pinned offline (Caffe is absent, the reference ships no
vectors), and both restatements were written by one author from one reading of the spec.  A shared misreading — a sign in
`dq = -1` / `data -= diff`, a normaliser, an Adam detail — would pass every parity test.  These tests therefore check what
the update must DO, against closed forms only; nothing here imports, links or executes anything under `oracle/`.

  (i)   actor ascent: with a hand-built critic Q(s, a) = k * a_j the actor's output j moves up for k > 0, down for k < 0, and
        avg_q rises in both cases (src/dqn.cpp:913-965); and a learned bandit, reward a closed-form function of one action
        parameter: the critic regresses onto it, mu(s) moves toward the optimum, avg_q rises
  (ii)  inverting gradients saturate: with mu_j at the bound the gradient pushes against, head row j receives exactly zero
        gradient and does not move (src/dqn.cpp:927-957)
  (iii) the TD target: beta = 1 -> y is the Monte-Carlo return; beta = 0, gamma = 0 -> y is the reward; terminal rows never
        bootstrap; and the critic regresses onto y (src/dqn.cpp:893-904)
  (iv)  SoftUpdateNet: tau = 1 -> the targets equal the online nets after one update, tau = 0 -> they never move; general tau:
        theta' = tau * theta + (1 - tau) * theta' element by element (src/dqn.cpp:1085-1096)
  (v)   ClipGradients: the gradient Adam applies has L2 norm <= clip_gradients, = clip when the raw norm exceeds it, and keeps
        the raw direction (src/dqn_main.cpp:35; Caffe SGDSolver::ClipGradients); first Adam step = lr * sign-ish step
"""
import numpy as np
import pytest

from synth import synth_states, synth_actions

pytestmark = pytest.mark.gpu

ACTOR, CRITIC, ACTOR_T, CRITIC_T = 0, 1, 2, 3
KIND_W, KIND_M, KIND_V, KIND_G = 0, 1, 2, 3
S = 20
HID = (64, 64)


# ---- dense (Caffe-order) parameter vectors: per layer W[N][K] then b[N]; actor heads W[4][H] b[4] W[6][H] b[6]; critic head W[1][H] b[1]
def layer_slices(S_in, hidden, heads):
    out, off, k = [], 0, S_in
    for n in tuple(hidden):
        out.append((off, off + n * k, n, k)); off += n * k
        out.append((off, off + n, n, 1)); off += n
        k = n
    for n in heads:
        out.append((off, off + n * k, n, k)); off += n * k
        out.append((off, off + n, n, 1)); off += n
    return out, off


def small_random(rng, count, scale=0.05):
    return (rng.standard_normal(count) * scale).astype(np.float32)


def linear_critic(S_, hidden, col, k):
    """Q(s, a) = k * lrelu^L(a[col]): one unit per layer carries the chosen input column through; everything else is zero."""
    sl, count = layer_slices(S_ + 10, hidden, (1,))
    w = np.zeros(count, np.float32)
    kin = S_ + 10
    for li in range(len(hidden)):
        a, b, n, kk = sl[2 * li]
        W = w[a:b].reshape(n, kk)
        W[0, (S_ + col) if li == 0 else 0] = 1.0
    a, b, n, kk = sl[2 * len(hidden)]
    w[a:b].reshape(n, kk)[0, 0] = k
    return w


def actor_with_bias(rng, S_, hidden, out_bias, scale=0.05):
    """small random tower, heads with small weights and the given output biases (10 values: 4 logits, 6 parameters)"""
    sl, count = layer_slices(S_, hidden, (4, 6))
    w = small_random(rng, count, scale)
    for i in range(len(hidden)):
        a, b, _, _ = sl[2 * i + 1]; w[a:b] = 0
    a, b, _, _ = sl[2 * len(hidden) + 1]; w[a:b] = out_bias[:4]
    a, b, _, _ = sl[2 * len(hidden) + 3]; w[a:b] = out_bias[4:]
    return w


def head_rows(S_, hidden):
    """-> list over the 10 actor outputs of (weight-row slice, bias index) in the dense vector"""
    sl, _ = layer_slices(S_, hidden, (4, 6))
    H = hidden[-1]
    rows = []
    a, _, _, _ = sl[2 * len(hidden)]; ba, _, _, _ = sl[2 * len(hidden) + 1]
    for j in range(4): rows.append((slice(a + j * H, a + (j + 1) * H), ba + j))
    a, _, _, _ = sl[2 * len(hidden) + 2]; ba, _, _, _ = sl[2 * len(hidden) + 3]
    for j in range(6): rows.append((slice(a + j * H, a + (j + 1) * H), ba + j))
    return rows


def fill_replay(dqn, rng, n, S_, reward_fn=None, actions=None, terminal_every=0, mc=None):
    s = synth_states(rng, n + 1, S_)
    a = synth_actions(rng, n) if actions is None else actions
    r = (rng.uniform(-1, 1, n) if reward_fn is None else reward_fn(s[:n], a)).astype(np.float32)
    term = np.zeros(n, np.uint8)
    if terminal_every:
        term[terminal_every - 1::terminal_every] = 1
    nx = s[1:].copy(); nx[term.astype(bool)] = 0
    mc = r.copy() if mc is None else mc.astype(np.float32)
    dqn.add_transitions_arrays(s[:n].copy(), a, r, mc, nx, term)
    return s[:n].copy(), a, r, mc, nx, term


def make(pkg, B=32, hidden=HID, **kw):
    kw.setdefault("memory", 4096)
    return pkg.DQN(S, minibatch=B, hidden=hidden, **kw)


# =========================================== (i) the actor ascends Q ===========================================================
@pytest.mark.parametrize("col,k,start", [(4, +0.5, 30.0), (4, -0.5, 30.0), (9, +0.2, 10.0), (9, -0.2, 10.0), (0, +3.0, 0.0), (0, -3.0, 0.0)])
def test_actor_moves_up_the_critics_gradient(pkg, gpu, col, k, start):
    """Hand-built, frozen critic Q = k * a[col] (for a[col] > 0; 0.01^L of it below): `BackwardFrom` with q diff = -1, the
    inverting-gradient scaling and `ApplyUpdate` together must move mu(s)[col] UP when k > 0 and DOWN when k < 0, and
    avg_q = mean Q(s, mu(s)) must rise either way.  A common-mode sign error anywhere in src/dqn.cpp:918-965 fails this."""
    rng = np.random.default_rng(5)
    B = 32
    # start > 0 keeps a[col] on the slope-1 side of every leaky ReLU of the carrier chain for the parameter outputs; the logit
    # case (col 0, start 0) sits at the kink: both sides have positive slope, the direction is still determined
    dqn = make(pkg, B=B, critic_lr=0.0, actor_lr=5e-3, tau=0.0, clip_grad=1e9, gamma=0.0, beta=0.0)
    bias = np.zeros(10, np.float32); bias[col] = start
    dqn.set_params(ACTOR, actor_with_bias(rng, S, HID, bias)); dqn.CloneNet(ACTOR)
    wc = linear_critic(S, HID, col, k)
    dqn.set_params(CRITIC, wc); dqn.CloneNet(CRITIC)
    states = fill_replay(dqn, rng, 512, S)[0]
    probe = states[:64]
    mu0 = dqn.SelectActionGreedily(probe)[:, col].mean()
    q0 = dqn.CriticForward(probe, dqn.SelectActionGreedily(probe)).mean()
    avg_q = []
    for u in range(60):
        idx = rng.integers(0, 512, B)
        _, aq = dqn.UpdateActorCritic(idx)
        avg_q.append(aq)
    mu1 = dqn.SelectActionGreedily(probe)[:, col].mean()
    q1 = dqn.CriticForward(probe, dqn.SelectActionGreedily(probe)).mean()
    assert np.array_equal(dqn.get_params(CRITIC), wc), "critic_lr = 0 must freeze the critic"
    moved = mu1 - mu0
    assert np.sign(moved) == np.sign(k) and abs(moved) > 0.05, (mu0, mu1)
    assert q1 > q0, (q0, q1)
    assert np.mean(avg_q[-10:]) > np.mean(avg_q[:10]), (avg_q[:3], avg_q[-3:])
    dqn.close()


def test_bandit_actor_finds_the_rewarded_parameter(pkg, gpu):
    """A one-step bandit: r = 1 - ((kick_power - 60) / 40)^2, gamma = 0, beta = 0 (so y = r), replay actions uniform in the
    parameter's range.  The critic must learn r(a) (its loss falls), and the actor, started at kick_power ~ 20, must move
    toward 60 while avg_q rises."""
    rng = np.random.default_rng(11)
    B, N, col, opt = 64, 2048, 8, 60.0      # col 8 = kick power (actionpara index 4: bounds [0, 100], src/dqn.cpp:942-943)
    dqn = make(pkg, B=B, hidden=(128, 128), critic_lr=2e-3, actor_lr=2e-3, tau=1.0, clip_grad=10.0, gamma=0.0, beta=0.0)
    bias = np.zeros(10, np.float32); bias[col] = 20.0
    dqn.set_params(ACTOR, actor_with_bias(rng, S, (128, 128), bias, scale=0.02)); dqn.CloneNet(ACTOR)
    sl, count = layer_slices(S + 10, (128, 128), (1,))
    wc = small_random(rng, count, 0.05)
    # the raw action parameters are O(100): scale the first layer's action columns so that pre-activations stay O(1)
    a0, b0, n0, k0 = sl[0]
    wc[a0:b0].reshape(n0, k0)[:, S + 4:] *= 0.02
    dqn.set_params(CRITIC, wc); dqn.CloneNet(CRITIC)
    acts = synth_actions(rng, N)
    acts[:, col] = rng.uniform(0, 100, N)
    reward = lambda s, a: 1.0 - ((a[:, col] - opt) / 40.0) ** 2
    states = fill_replay(dqn, rng, N, S, reward_fn=reward, actions=acts)[0]
    probe = states[:128]
    mu0 = dqn.SelectActionGreedily(probe)[:, col].mean()
    loss, avg_q = [], []
    for u in range(600):
        l, aq = dqn.UpdateActorCritic(rng.integers(0, N, B))
        loss.append(l); avg_q.append(aq)
    mu1 = dqn.SelectActionGreedily(probe)[:, col].mean()
    assert np.mean(loss[-50:]) < 0.25 * np.mean(loss[:20]), (np.mean(loss[:20]), np.mean(loss[-50:]))
    # the critic's picture of the bandit: better at the optimum than 40 away from it, on both sides
    mid = synth_actions(rng, 128); lo = mid.copy(); hi = mid.copy()
    mid[:, col] = opt; lo[:, col] = opt - 40; hi[:, col] = opt + 40
    qm, ql, qh = (dqn.CriticForward(probe, a).mean() for a in (mid, lo, hi))
    assert qm > ql + 0.3 and qm > qh + 0.3, (ql, qm, qh)
    assert abs(mu1 - opt) < abs(mu0 - opt) - 5.0 and mu1 > mu0, (mu0, mu1)
    assert np.mean(avg_q[-50:]) > np.mean(avg_q[100:150]), (np.mean(avg_q[100:150]), np.mean(avg_q[-50:]))
    dqn.close()


# =========================================== (ii) inverting gradients saturate ==================================================
@pytest.mark.parametrize("col,bound,k", [(4, 100.0, +0.5), (8, 100.0, +1.0), (4, 0.0, -0.5), (5, 180.0, +0.3), (5, -180.0, -0.3),
                                           (9, -180.0, -0.2), (0, 1.0, +2.0), (3, -1.0, -2.0)])
def test_output_at_its_bound_gets_no_gradient(pkg, gpu, col, bound, k):
    """mu_j == max and the critic wants more of it (or == min and it wants less): the inverting-gradient factor
    (max - output) / (max - min) [resp. (output - min) / ...] is exactly 0 (src/dqn.cpp:927-957), so head row j's gradient, its
    Adam history and its weights stay exactly where they were — while the other rows do move."""
    rng = np.random.default_rng(3)
    B = 32
    dqn = make(pkg, B=B, critic_lr=0.0, actor_lr=1e-2, tau=0.0, clip_grad=1e9, gamma=0.0, beta=0.0)
    rows = head_rows(S, HID)
    bias = rng.uniform(-0.5, 0.5, 10).astype(np.float32)
    bias[4] = 50; bias[8] = 50
    wa = actor_with_bias(rng, S, HID, bias)
    wa[rows[col][0]] = 0.0; wa[rows[col][1]] = bound          # output j is the bound for every state, exactly
    dqn.set_params(ACTOR, wa); dqn.CloneNet(ACTOR)
    # the critic likes output `col` (sign k) and, through a second carrier unit, output `other` too: that row must move
    other = 6 if col != 6 else 7
    sl, count = layer_slices(S + 10, HID, (1,))
    wc = linear_critic(S, HID, col, k)
    for li in range(len(HID)):
        a, b, n, kk = sl[2 * li]
        wc[a:b].reshape(n, kk)[1, (S + other) if li == 0 else 1] = 1.0
        if li == 0: wc[sl[1][0] + 1] = 400.0                  # bias: keeps the second carrier on the slope-1 side
    a, b, n, kk = sl[2 * len(HID)]
    wc[a:b].reshape(n, kk)[0, 1] = 0.25
    dqn.set_params(CRITIC, wc); dqn.CloneNet(CRITIC)
    fill_replay(dqn, rng, 256, S)
    before = dqn.get_params(ACTOR)
    for u in range(3):
        dqn.UpdateActorCritic(rng.integers(0, 256, B))
        out = dqn.debug_read("actor_out")
        assert np.all(out[:, col] == np.float32(bound))
        dq = dqn.debug_read("dq_da")
        assert np.all(dq[:, col] == 0.0), dq[:4, col]
        assert np.any(dq[:, other] != 0.0)
    after = dqn.get_params(ACTOR)
    g, m, v = (dqn.get_params(ACTOR, kind) for kind in (KIND_G, KIND_M, KIND_V))
    wr, br = rows[col]
    assert np.all(g[wr] == 0) and g[br] == 0 and np.all(m[wr] == 0) and m[br] == 0 and np.all(v[wr] == 0) and v[br] == 0
    assert np.array_equal(after[wr], before[wr]) and after[br] == before[br]
    wo, bo = rows[other]
    assert after[bo] != before[bo] and np.any(after[wo] != before[wo])
    dqn.close()


def test_inverting_gradient_scales_with_the_headroom(pkg, gpu):
    """Same critic slope, three starting points of dash power (bounds [0, 100]): the post-invert gradient is the raw one
    times (100 - p) / 100 for an 'increase' gradient — so 10 -> 0.9, 50 -> 0.5, 90 -> 0.1 of the raw slope, in ratio."""
    rng = np.random.default_rng(4)
    B, col, k = 32, 4, 0.5
    got = []
    for p in (10.0, 50.0, 90.0):
        dqn = make(pkg, B=B, critic_lr=0.0, actor_lr=0.0, tau=0.0, clip_grad=1e9)
        bias = np.zeros(10, np.float32)
        wa = actor_with_bias(rng, S, HID, bias)
        rows = head_rows(S, HID)
        wa[rows[col][0]] = 0.0; wa[rows[col][1]] = p
        dqn.set_params(ACTOR, wa); dqn.CloneNet(ACTOR)
        dqn.set_params(CRITIC, linear_critic(S, HID, col, k)); dqn.CloneNet(CRITIC)
        fill_replay(dqn, rng, 128, S)
        dqn.UpdateActorCritic(rng.integers(0, 128, B))
        dq = dqn.debug_read("dq_da")[:, col]
        assert np.all(dq == dq[0])
        got.append(float(dq[0]))
        dqn.close()
    # raw diff = dLoss/da with q diff = -1  ->  -k ; 'diff < 0' -> times (max - output) / (max - min)
    np.testing.assert_allclose(got, [-k * 0.9, -k * 0.5, -k * 0.1], rtol=1e-6)


# =========================================== (iii) the TD target ===============================================================
def test_td_target_limits(pkg, gpu):
    rng = np.random.default_rng(7)
    B, N = 32, 256
    for beta, gamma in ((1.0, 0.99), (0.0, 0.0), (0.0, 0.5), (0.5, 0.9)):
        dqn = make(pkg, B=B, beta=beta, gamma=gamma, critic_lr=1e-3, actor_lr=1e-5)
        for net, sz in ((ACTOR, S), (CRITIC, S + 10)):
            sl, count = layer_slices(sz, HID, (4, 6) if net == ACTOR else (1,))
            w = small_random(rng, count, 0.1)
            if net == CRITIC: w[sl[0][0]:sl[0][1]].reshape(HID[0], S + 10)[:, S + 4:] *= 0.02
            dqn.set_params(net, w); dqn.CloneNet(net)
        mcs = rng.uniform(-3, 3, N)
        s, a, r, mc, nx, term = fill_replay(dqn, rng, N, S, terminal_every=5, mc=mcs)
        idx = rng.integers(0, N, B)
        dqn.UpdateActorCritic(idx)
        y, qt, t = dqn.debug_read("y"), dqn.debug_read("q_target"), term[idx].astype(bool)
        assert np.array_equal(dqn.debug_read("terminal") != 0, t)
        if beta == 1.0:
            np.testing.assert_array_equal(y, mc[idx])                         # Monte-Carlo return only (src/dqn.cpp:897)
        elif gamma == 0.0:
            np.testing.assert_array_equal(y, r[idx])                          # reward only
        else:
            off = np.where(t, r[idx].astype(np.float64), r[idx].astype(np.float64) + gamma * qt.astype(np.float64))
            want = beta * mc[idx].astype(np.float64) + (1 - beta) * off
            np.testing.assert_allclose(y, want, rtol=1e-6, atol=1e-6)
            assert np.any(np.abs(qt[~t]) > 1e-4), "the non-terminal rows must have bootstrapped from a non-trivial Q'"
        dqn.close()


def test_critic_regresses_onto_the_target(pkg, gpu):
    """beta = 1: the critic's only teacher is the Monte-Carlo return, here a closed-form function of the state — repeated
    Step(1)s must drive Q(s, a) onto it (the EuclideanLoss gradient has the descent sign), whatever the actor does."""
    rng = np.random.default_rng(8)
    B, N = 64, 1024
    dqn = make(pkg, B=B, hidden=(128, 128), beta=1.0, critic_lr=2e-3, actor_lr=0.0, tau=0.01)
    for net, sz in ((ACTOR, S), (CRITIC, S + 10)):
        sl, count = layer_slices(sz, (128, 128), (4, 6) if net == ACTOR else (1,))
        w = small_random(rng, count, 0.05)
        if net == CRITIC: w[sl[0][0]:sl[0][1]].reshape(128, S + 10)[:, S + 4:] *= 0.02
        dqn.set_params(net, w); dqn.CloneNet(net)
    st = synth_states(rng, N + 1, S)
    mc = 2.0 * st[:N, 0] - st[:N, 3] + 0.5
    s, a, r, mc, nx, term = fill_replay(dqn, np.random.default_rng(8), N, S, mc=mc)
    mc = (2.0 * s[:, 0] - s[:, 3] + 0.5).astype(np.float32)
    dqn.ClearReplayMemory()
    dqn.add_transitions_arrays(s, a, r, mc, nx, term)
    err0 = np.mean((dqn.CriticForward(s[:256], a[:256]) - mc[:256]) ** 2)
    losses = [dqn.UpdateActorCritic(rng.integers(0, N, B))[0] for _ in range(500)]
    err1 = np.mean((dqn.CriticForward(s[:256], a[:256]) - mc[:256]) ** 2)
    assert err1 < 0.1 * err0, (err0, err1)
    assert np.mean(losses[-50:]) < 0.1 * np.mean(losses[:10])
    # the reported loss IS sum (q - y)^2 / (2 B) of that minibatch (Caffe EuclideanLoss)
    idx = rng.integers(0, N, B)
    q_before = dqn.CriticForward(s[idx], a[idx]).astype(np.float64)
    l, _ = dqn.UpdateActorCritic(idx)
    np.testing.assert_allclose(l, np.sum((q_before - mc[idx]) ** 2) / (2 * B), rtol=1e-4)
    dqn.close()


# =========================================== (iv) SoftUpdateNet ================================================================
@pytest.mark.parametrize("tau", [1.0, 0.0, 0.25])
def test_soft_update(pkg, gpu, tau):
    rng = np.random.default_rng(9)
    B = 32
    dqn = make(pkg, B=B, tau=tau, critic_lr=1e-2, actor_lr=1e-2)
    for net, sz in ((ACTOR, S), (CRITIC, S + 10)):
        _, count = layer_slices(sz, HID, (4, 6) if net == ACTOR else (1,))
        dqn.set_params(net, small_random(rng, count, 0.1))
        dqn.set_params(net + 2, small_random(rng, count, 0.1))     # targets deliberately different from the online nets
    fill_replay(dqn, rng, 256, S)
    t0 = [dqn.get_params(n) for n in (ACTOR_T, CRITIC_T)]
    for u in range(2):
        dqn.UpdateActorCritic(rng.integers(0, 256, B))
        w = [dqn.get_params(n) for n in (ACTOR, CRITIC)]
        t1 = [dqn.get_params(n) for n in (ACTOR_T, CRITIC_T)]
        for k in range(2):
            if tau == 1.0:
                np.testing.assert_array_equal(t1[k], w[k])
            elif tau == 0.0:
                np.testing.assert_array_equal(t1[k], t0[k])
            else:
                want = tau * w[k].astype(np.float64) + (1 - tau) * t0[k].astype(np.float64)
                np.testing.assert_allclose(t1[k], want, rtol=2e-7, atol=1e-7)
                assert not np.array_equal(t1[k], t0[k])
        t0 = t1
    dqn.close()


# =========================================== (v) ClipGradients + the first Adam step ===========================================
@pytest.mark.parametrize("clip", [0.5, 10.0, 1e6])
def test_applied_gradient_is_clipped(pkg, gpu, clip):
    """After the FIRST Adam step from zero history, m = (1 - beta1) * g_applied, v = (1 - beta2) * g_applied^2: the applied
    gradient is read back from m.  Its norm is min(raw norm, clip); its direction is the raw gradient's; and the first step
    moves every touched weight by lr * g / (|g| + eps * sqrt(1 - beta2) / ...) ~ lr in the descent direction."""
    rng = np.random.default_rng(10)
    B, b1, b2, lr = 32, 0.95, 0.999, 1e-3
    dqn = make(pkg, B=B, clip_grad=clip, critic_lr=lr, actor_lr=lr, momentum=b1, momentum2=b2, tau=0.0)
    for net, sz in ((ACTOR, S), (CRITIC, S + 10)):
        sl, count = layer_slices(sz, HID, (4, 6) if net == ACTOR else (1,))
        w = small_random(rng, count, 0.3)
        if net == CRITIC: w[sl[0][0]:sl[0][1]].reshape(HID[0], S + 10)[:, S + 4:] *= 0.02
        dqn.set_params(net, w); dqn.CloneNet(net)
    fill_replay(dqn, rng, 256, S, reward_fn=lambda s, a: 20.0 * s[:, 0])
    w0 = [dqn.get_params(n) for n in (ACTOR, CRITIC)]
    dqn.UpdateActorCritic(rng.integers(0, 256, B))
    for net in (CRITIC, ACTOR):
        g = dqn.get_params(net, KIND_G).astype(np.float64)
        m = dqn.get_params(net, KIND_M).astype(np.float64)
        v = dqn.get_params(net, KIND_V).astype(np.float64)
        w1 = dqn.get_params(net).astype(np.float64)
        applied = m / (1 - b1)
        raw_norm, app_norm = np.linalg.norm(g), np.linalg.norm(applied)
        assert raw_norm > 0
        assert app_norm <= clip * (1 + 1e-5), (raw_norm, app_norm, clip)
        np.testing.assert_allclose(app_norm, min(raw_norm, clip), rtol=1e-4)
        if clip == 0.5:
            assert raw_norm > clip, "this case is meant to have the clip active"
        cos = np.dot(applied, g) / (app_norm * raw_norm)
        assert cos > 1 - 1e-6
        np.testing.assert_allclose(v, (1 - b2) * applied ** 2, rtol=1e-4, atol=1e-30)
        # first step of Caffe's AdamSolver (t = 1; update = lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps), eps NOT bias-corrected)
        #   = lr * g / (|g| + eps / sqrt(1 - b2)): every weight whose gradient is well above 3e-7 moves by lr against its sign
        step = w1 - w0[net].astype(np.float64)
        big = np.abs(applied) > 3e-3
        assert big.sum() > 20
        np.testing.assert_allclose(step[big], -lr * np.sign(applied[big]), rtol=1e-3)
        np.testing.assert_allclose(step, -lr * applied / (np.abs(applied) + 1e-8 / np.sqrt(1 - b2)), rtol=2e-3, atol=2e-8)
        assert np.all(step[applied == 0] == 0)
    dqn.close()


# =========================================== the fp16-MFMA learner behaves the same way ========================================
@pytest.mark.parametrize("k", [+0.5, -0.5])
def test_fp16_learner_moves_up_the_critics_gradient(pkg, gpu, k):
    """BASELINE configs[4]'s precision (fp16 MFMA operands, fp32 accumulate; master weights, heads, Adam in fp32): against the
    fp32 oracle its gradients are only 1-2 % tight (tests/test_gpu_fp16.py), so the direction test matters more here.  Hand-built
    frozen critic Q = k * dash_power: mu(s)[4] must move with the sign of k and avg_q must rise."""
    rng = np.random.default_rng(6)
    B, hid, col = 128, (128, 128), 4
    dqn = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, precision="fp16", critic_lr=0.0, actor_lr=5e-3, tau=0.0, clip_grad=1e9,
                  gamma=0.0, beta=0.0)
    bias = np.zeros(10, np.float32); bias[col] = 30.0
    dqn.set_params(ACTOR, actor_with_bias(rng, S, hid, bias)); dqn.CloneNet(ACTOR)
    dqn.set_params(CRITIC, linear_critic(S, hid, col, k)); dqn.CloneNet(CRITIC)
    states = fill_replay(dqn, rng, 1024, S)[0]
    probe = states[:64]
    mu0 = dqn.SelectActionGreedily(probe)[:, col].mean()
    avg_q = [dqn.UpdateActorCritic(rng.integers(0, 1024, B))[1] for _ in range(60)]
    mu1 = dqn.SelectActionGreedily(probe)[:, col].mean()
    assert np.sign(mu1 - mu0) == np.sign(k) and abs(mu1 - mu0) > 0.05, (mu0, mu1)
    assert np.mean(avg_q[-10:]) > np.mean(avg_q[:10]), (avg_q[:3], avg_q[-3:])
    dqn.close()


def test_fp16_learner_bandit(pkg, gpu):
    """The one-step bandit of test_bandit_actor_finds_the_rewarded_parameter on the fp16 learner."""
    rng = np.random.default_rng(12)
    B, N, col, opt, hid = 128, 2048, 8, 60.0, (128, 128)
    dqn = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, precision="fp16", critic_lr=2e-3, actor_lr=2e-3, tau=1.0, clip_grad=10.0,
                  gamma=0.0, beta=0.0)
    bias = np.zeros(10, np.float32); bias[col] = 20.0
    dqn.set_params(ACTOR, actor_with_bias(rng, S, hid, bias, scale=0.02)); dqn.CloneNet(ACTOR)
    sl, count = layer_slices(S + 10, hid, (1,))
    wc = small_random(rng, count, 0.05)
    a0, b0, n0, k0 = sl[0]
    wc[a0:b0].reshape(n0, k0)[:, S + 4:] *= 0.02
    dqn.set_params(CRITIC, wc); dqn.CloneNet(CRITIC)
    acts = synth_actions(rng, N); acts[:, col] = rng.uniform(0, 100, N)
    states = fill_replay(dqn, rng, N, S, reward_fn=lambda s, a: 1.0 - ((a[:, col] - opt) / 40.0) ** 2, actions=acts)[0]
    probe = states[:128]
    mu0 = dqn.SelectActionGreedily(probe)[:, col].mean()
    loss, avg_q = [], []
    for u in range(400):
        l, aq = dqn.UpdateActorCritic(rng.integers(0, N, B)); loss.append(l); avg_q.append(aq)
    mu1 = dqn.SelectActionGreedily(probe)[:, col].mean()
    assert np.mean(loss[-50:]) < 0.3 * np.mean(loss[:20]), (np.mean(loss[:20]), np.mean(loss[-50:]))
    assert abs(mu1 - opt) < abs(mu0 - opt) - 5.0 and mu1 > mu0, (mu0, mu1)
    assert np.mean(avg_q[-50:]) > np.mean(avg_q[60:110]), (np.mean(avg_q[60:110]), np.mean(avg_q[-50:]))
    dqn.close()


# =========================================== normalisers and ordering =================================
def test_gradient_normalisers_and_the_order_of_the_two_steps(pkg, gpu):
    """From the library's own per-row outputs, no oracle: (5) the critic's loss gradient is AVERAGED over the minibatch
    (EuclideanLoss: d loss / d q_m = (q_m - y_m) / B, so the q head's bias gradient is mean(q - y)) while the actor's gradient is
    a SUM over rows (q diff = -1 per row, src/dqn.cpp:918-921: the head bias gradients are the column sums of the post-invert
    diffs); (6) avg_q is evaluated with the critic AFTER its step and the policy BEFORE the actor's (src/dqn.cpp:904-916)."""
    rng = np.random.default_rng(13)
    B, N = 32, 512
    dqn = make(pkg, B=B, critic_lr=5e-3, actor_lr=5e-3, tau=0.5, clip_grad=1e9)
    for net, sz in ((ACTOR, S), (CRITIC, S + 10)):
        sl, count = layer_slices(sz, HID, (4, 6) if net == ACTOR else (1,))
        w = small_random(rng, count, 0.2)
        if net == CRITIC: w[sl[0][0]:sl[0][1]].reshape(HID[0], S + 10)[:, S + 4:] *= 0.02
        dqn.set_params(net, w); dqn.CloneNet(net)
    s, a, r, mc, nx, term = fill_replay(dqn, rng, N, S, terminal_every=7)
    rows = head_rows(S, HID)
    slc, _ = layer_slices(S + 10, HID, (1,))
    q_bias = slc[2 * len(HID) + 1][0]
    for u in range(3):
        idx = rng.integers(0, N, B)
        mu_before = dqn.SelectActionGreedily(s[idx])                       # the policy this update's avg_q must use
        loss, avg_q = dqn.UpdateActorCritic(idx)
        q, y = dqn.debug_read("q_train").astype(np.float64), dqn.debug_read("y").astype(np.float64)
        gc, ga = dqn.get_params(CRITIC, KIND_G).astype(np.float64), dqn.get_params(ACTOR, KIND_G).astype(np.float64)
        np.testing.assert_allclose(gc[q_bias], np.mean(q - y), rtol=1e-5, atol=1e-7)          # averaged over B
        dq = dqn.debug_read("dq_da").astype(np.float64)
        for j in range(10):
            np.testing.assert_allclose(ga[rows[j][1]], dq[:, j].sum(), rtol=1e-5, atol=1e-6)    # summed over B
        np.testing.assert_allclose(dqn.debug_read("actor_out"), mu_before, rtol=1e-5, atol=1e-5)
        q_after = dqn.CriticForward(s[idx], mu_before)                      # the critic has taken its step, the policy is the old one
        np.testing.assert_allclose(avg_q, q_after.astype(np.float64).mean(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dqn.debug_read("q_policy"), q_after, rtol=1e-5, atol=1e-6)
        assert not np.allclose(dqn.CriticForward(s[idx], dqn.SelectActionGreedily(s[idx])), q_after, rtol=1e-5, atol=1e-6), "the actor did move"
    dqn.close()
