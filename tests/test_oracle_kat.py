"""Hand-derived known-answer tests of the CPU restatement (oracle/dqn_oracle.c): every value
below is computed by hand / plain numpy from the reference's lines, not by the code under test."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle.c_oracle import Oracle, GameState, get_action, label_transitions


def test_get_action_argmax_and_params():
    # src/dqn.cpp:196-208 — TACKLE (idx 2) is masked, first maximum wins, params by GetParamOffset
    ao = np.array([[0.1, 0.5, 9.0, 0.2, 10, 11, 12, 13, 14, 15],      # TURN wins although tackle logit is 9
                   [0.7, 0.7, 0.0, 0.7, 10, 11, 12, 13, 14, 15],      # 3-way tie -> DASH (lowest index)
                   [-1, -1, 5.0, -0.5, 10, 11, 12, 13, 14, 15],       # KICK
                   [-99998, -99998.5, 0, -99998.2, 1, 2, 3, 4, 5, 6]],  # all above the -99999 mask -> DASH
                  np.float32)
    act, a1, a2 = get_action(ao)
    assert list(act) == [1, 0, 3, 0]
    assert list(a1) == [12, 10, 14, 1]          # TURN->params[2]; DASH->params[0]; KICK->params[4]
    assert list(a2) == [0, 11, 15, 2]           # TURN has no second arg; DASH->[1]; KICK->[5]


def test_label_transitions():
    # src/dqn.cpp:783-797 with gamma = .5 : [1,2,4] -> [1+.5*(2+.5*4), 2+.5*4, 4] = [3,4,4]
    np.testing.assert_array_equal(label_transitions(0.5, [1, 2, 4]), np.array([3, 4, 4], np.float32))
    # double intermediate, float store: 0.1f + 0.99*0.2f
    r = np.array([0.1, 0.2], np.float32)
    exp0 = np.float32(np.float64(r[0]) + 0.99 * np.float64(r[1]))
    out = label_transitions(0.99, r)
    assert out[1] == r[1] and out[0] == exp0


def test_add_transitions_eviction_counts():
    # src/dqn.cpp:768-781: AddTransitions pops while size+n >= cap (so <= cap-1 remain);
    # AddTransition pops iff size == cap (so cap remain)
    S, cap = 3, 10
    o = Oracle(B=32, S=S, hidden=(64,), capacity=cap)
    mk = lambda n, base: (np.full((n, S), base, np.float32) + np.arange(n, dtype=np.float32)[:, None],
                          np.zeros((n, 10), np.float32), np.arange(n, dtype=np.float32) + base,
                          np.zeros(n, np.float32), np.zeros((n, S), np.float32), np.zeros(n, np.uint8))
    o.add_transitions(*mk(4, 100)); assert o.memory_size() == 4
    o.add_transitions(*mk(4, 200)); assert o.memory_size() == 8
    o.add_transitions(*mk(4, 300))                     # 8+4 >= 10 -> pop until size+4 < 10 -> size 5 -> 9
    assert o.memory_size() == 9
    r = o.read_memory(0, 9)[2]
    np.testing.assert_array_equal(r, [103, 200, 201, 202, 203, 300, 301, 302, 303])
    z = np.zeros(S, np.float32)
    o.add_transition(z, np.zeros(10, np.float32), 7.0, 0.0, z, 0); assert o.memory_size() == 10   # fills to cap
    o.add_transition(z, np.zeros(10, np.float32), 8.0, 0.0, z, 0); assert o.memory_size() == 10   # evicts one
    r = o.read_memory(0, 10)[2]
    assert r[0] == 200 and r[-1] == 8 and r[-2] == 7
    o.clear_memory(); assert o.memory_size() == 0
    o.close()


def _tiny(B=32, S=2, H=64, **kw):
    return Oracle(B=B, S=S, hidden=(H,), capacity=64, **kw)


def test_forward_by_hand():
    # actor: one hidden layer; W1 = e-rows, check leaky slope 0.01 and head wiring
    S, H = 2, 64
    o = _tiny(S=S, H=H)
    n = o.param_count(0)
    w = np.zeros(n, np.float32)
    W1 = np.zeros((H, S), np.float32); W1[0] = [1, 0]; W1[1] = [0, -2]
    b1 = np.zeros(H, np.float32); b1[2] = 0.5
    Wa = np.zeros((4, H), np.float32); Wa[0, 0] = 1; Wa[1, 1] = 1; Wa[3, 2] = 2
    ba = np.array([0, 0, 0, 1], np.float32)
    Wp = np.zeros((6, H), np.float32); Wp[4, 0] = 10
    bp = np.arange(6, dtype=np.float32)
    w[:] = np.concatenate([W1.ravel(), b1, Wa.ravel(), ba, Wp.ravel(), bp])
    o.set_params(0, w)
    out = o.actor_forward(np.array([[3.0, 4.0], [-3.0, -4.0]], np.float32))
    # row0: h = [3, lrelu(-8) = -0.08, 0.5, 0...] -> actions [3, -0.08, 0, 2*0.5+1], params [0,1,2,3, 4+30, 5]
    np.testing.assert_allclose(out[0], [3, -0.08, 0, 2, 0, 1, 2, 3, 34, 5], rtol=1e-6)
    # row1: h = [lrelu(-3) = -0.03, 8, 0.5]
    np.testing.assert_allclose(out[1], [-0.03, 8, 0, 2, 0, 1, 2, 3, 4 - 0.3, 5], rtol=1e-6)
    o.close()


def test_td_target_and_loss_by_hand():
    # critic with all-zero weights except the head bias -> Q == bq everywhere, so
    #   y = beta*mc + (1-beta)*(term ? r : r + gamma*bq_target) ; loss = sum((bq - y)^2) / (2B)
    B, S = 32, 2
    o = _tiny(B=B, S=S, gamma=0.9, beta=0.25)
    wc = np.zeros(o.param_count(1), np.float32); wc[-1] = 2.0
    o.set_params(1, wc); o.clone_to_target(1)
    wt = wc.copy(); wt[-1] = 4.0
    o.set_params(3, wt)                                   # target critic: Q' == 4
    o.set_params(0, np.zeros(o.param_count(0), np.float32)); o.clone_to_target(0)
    s = np.zeros((B, S), np.float32); a = np.zeros((B, 10), np.float32)
    r = np.linspace(-1, 1, B).astype(np.float32); mc = np.full(B, 8.0, np.float32)
    term = (np.arange(B) % 2).astype(np.uint8)
    o.add_transitions(s, a, r, mc, s, term)
    loss, avgq = o.update(np.arange(B))
    y = np.where(term == 1, 0.25 * 8 + 0.75 * r, 0.25 * 8 + 0.75 * (r + 0.9 * 4.0))
    np.testing.assert_allclose(o.debug_read("y"), y, rtol=1e-6)
    np.testing.assert_allclose(o.debug_read("q_train"), 2.0)
    assert abs(loss - ((2.0 - y) ** 2).sum() / (2 * B)) < 1e-5
    o.close()


def test_adam_first_step_by_hand():
    # With only the head bias free (all activations zero), the critic gradient is
    #   d loss / d bq = sum(q - y)/B =: g ; Adam t=1: m=(1-b1)g, v=(1-b2)g^2,
    #   step = lr*sqrt(1-b2)/(1-b1) * m/(sqrt(v)+eps) -> lr * g/(|g| + eps/sqrt(1-b2))
    B, S = 32, 2
    o = _tiny(B=B, S=S, gamma=0.9, beta=0.0, lr_critic=1e-3, clip=1e9)
    wc = np.zeros(o.param_count(1), np.float32); wc[-1] = 2.0
    o.set_params(1, wc); o.clone_to_target(1)
    o.set_params(0, np.zeros(o.param_count(0), np.float32)); o.clone_to_target(0)
    s = np.zeros((B, S), np.float32); a = np.zeros((B, 10), np.float32)
    r = np.full(B, 1.0, np.float32)
    o.add_transitions(s, a, r, r, s, np.ones(B, np.uint8))       # all terminal: y = r = 1
    o.update(np.arange(B))
    g = (2.0 - 1.0)                                       # mean over B of (q - y)
    m, v = 0.05 * g, 0.001 * g * g
    corr = np.sqrt(1 - 0.999) / (1 - 0.95)
    step = 1e-3 * corr * m / (np.sqrt(v) + 1e-8)
    assert abs(o.get_params(1)[-1] - (2.0 - step)) < 1e-7
    assert abs(o.get_params(1, 1)[-1] - m) < 1e-7 and abs(o.get_params(1, 2)[-1] - v) < 1e-7   # (1-0.95f), (1-0.999f) are not exact
    # soft update (tau=.001): target = .001*new + .999*2.0
    assert abs(o.get_params(3)[-1] - (0.001 * (2.0 - step) + 0.999 * 2.0)) < 1e-6
    assert o.get_iters() == (1, 1)
    o.close()


def test_clip_gradients_by_hand():
    # same setup, clip = 0.25 < |g| = 1 -> g scaled to 0.25 before Adam
    B, S = 32, 2
    o = _tiny(B=B, S=S, beta=0.0, clip=0.25)
    wc = np.zeros(o.param_count(1), np.float32); wc[-1] = 2.0
    o.set_params(1, wc); o.clone_to_target(1)
    o.set_params(0, np.zeros(o.param_count(0), np.float32)); o.clone_to_target(0)
    s = np.zeros((B, S), np.float32); a = np.zeros((B, 10), np.float32); r = np.ones(B, np.float32)
    o.add_transitions(s, a, r, r, s, np.ones(B, np.uint8))
    o.update(np.arange(B))
    assert abs(o.get_params(1, 1)[-1] - 0.05 * 0.25) < 1e-7          # m = (1-b1) * clipped g
    o.close()


@pytest.mark.parametrize("diff,out,mn,mx,exp", [
    (-2.0, 0.5, -1, 1, -2.0 * (1 - 0.5) / 2),       # wants to increase: scale by (max-out)/(max-min)
    (+2.0, 0.5, -1, 1, +2.0 * (0.5 + 1) / 2),       # wants to decrease: scale by (out-min)/(max-min)
    (0.0, 0.5, -1, 1, 0.0),
    (-1.0, 150.0, 0, 100, -1.0 * (100 - 150) / 100),  # beyond the bound: sign flips (inverting!)
])
def test_inverting_gradients_formula(diff, out, mn, mx, exp):
    # src/dqn.cpp:927-957, evaluated in float like the reference
    d, o_, mn_, mx_ = np.float32(diff), np.float32(out), np.float32(mn), np.float32(mx)
    if d < 0:
        d = d * ((mx_ - o_) / (mx_ - mn_))
    elif d > 0:
        d = d * ((o_ - mn_) / (mx_ - mn_))
    assert abs(float(d) - exp) < 1e-6


def test_game_state_reward_by_hand():
    # src/hfo_game.cpp:122-236.  Two steps; ball proximity 0.2 -> 0.5, kickable -1 -> +1,
    # we are on the ball (unum 7): reward = d(ball_prox) + 1 (first kickable) + 3*(-d dist_goal)
    def state(ball_prox, goal_prox, kick, ball_th, goal_th):
        s = np.zeros(59, np.float32)
        s[53], s[15], s[12] = ball_prox, goal_prox, kick
        s[51], s[52] = np.sin(ball_th), np.cos(ball_th)
        s[13], s[14] = np.sin(goal_th), np.cos(goal_th)
        return s

    def dist(bp, gp, bt, gt):
        bd, gd = 1 - bp, 1 - gp
        return np.sqrt(bd * bd + gd * gd - 2 * bd * gd * np.cos(abs(bt - gt)))

    g = GameState(unum=7)
    g.update(state(0.2, 0.3, -1, 0.3, -0.4), status=0, player_on_ball=7)
    assert abs(g.reward() - 0.0) < 1e-6                  # first step: all deltas are 0
    g.update(state(0.5, 0.35, +1, 0.1, -0.2), status=0, player_on_ball=7)
    d0, d1 = dist(0.2, 0.3, 0.3, -0.4), dist(0.5, 0.35, 0.1, -0.2)
    exp = (0.5 - 0.2) + 1.0 + 3.0 * -(d1 - d0)
    assert abs(g.reward() - exp) < 1e-5
    # goal by us: +5, deltas zeroed because the episode is over (:164-168)
    g.update(state(0.9, 0.9, +1, 0.0, 0.0), status=1, player_on_ball=7)
    assert abs(g.reward() - 5.0) < 1e-6
