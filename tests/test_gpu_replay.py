"""Device-resident replay ring vs the reference's std::deque semantics (oracle model), the
acting-time API and the error convention."""
import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle
from synth import synth_actions, synth_replay, synth_states

pytestmark = pytest.mark.gpu


def _mk(rng, n, S):
    return synth_replay(rng, n, S, mean_len=5)


def test_ring_eviction_matches_deque(pkg, gpu):
    """AddTransitions keeps <= cap-1, AddTransition keeps <= cap (src/dqn.cpp:768-781); logical
    order survives wrap-around."""
    S, cap = 59, 100
    rng = np.random.default_rng(0)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=cap)
    orc = c_oracle.Oracle(B=32, S=S, hidden=(64,), capacity=cap)
    for n in (30, 30, 30, 30, 45, 7, 99, 1, 64):
        d = _mk(rng, n, S)
        dqn.add_transitions_arrays(*d); orc.add_transitions(*d)
        assert dqn.memory_size() == orc.memory_size() <= cap - 1
        a, b = dqn.read_memory(0, dqn.memory_size()), orc.read_memory(0, orc.memory_size())
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    for i in range(150):                              # single adds: fill to cap, then FIFO evict
        d = _mk(rng, 1, S)
        t = pkg.Transition(d[0][0], d[1][0], float(d[2][0]), float(d[3][0]), None if d[5][0] else d[4][0])
        dqn.AddTransition(t)
        orc.add_transition(d[0][0], d[1][0], d[2][0], d[3][0], d[4][0], d[5][0])
        assert dqn.memory_size() == orc.memory_size()
    assert dqn.memory_size() == cap
    a, b = dqn.read_memory(0, cap), orc.read_memory(0, cap)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    # AddTransitions of an EMPTY vector on a full deque: `while (size() + 0 >= capacity) pop_front()` (src/dqn.cpp:776) evicts the
    # oldest transition and appends nothing; on a deque below capacity it is a no-op
    dqn.AddTransitions([])
    assert dqn.memory_size() == cap - 1
    a2 = dqn.read_memory(0, cap - 1)
    for x, y in zip(a2, a):
        np.testing.assert_array_equal(x, y[1:])
    dqn.AddTransitions([])
    assert dqn.memory_size() == cap - 1
    dqn.ClearReplayMemory(); assert dqn.memory_size() == 0
    dqn.AddTransitions([]); assert dqn.memory_size() == 0
    dqn.close(); orc.close()


def test_transition_objects_and_label(pkg, gpu):
    """The reference's calling sequence: LabelTransitions then AddTransitions on a list of
    Transition tuples (src/dqn_main.cpp:145-150)."""
    S = 59
    rng = np.random.default_rng(1)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=1000, gamma=0.9)
    st = synth_states(rng, 11, S); ac = synth_actions(rng, 10)
    ep = [pkg.Transition(st[i], ac[i], float(i), 0.0, st[i + 1] if i < 9 else None) for i in range(10)]
    ep = dqn.LabelTransitions(ep)
    exp = c_oracle.label_transitions(0.9, np.arange(10, dtype=np.float32))
    np.testing.assert_array_equal([t.on_policy_target for t in ep], exp)
    dqn.AddTransitions(ep)
    s, a, r, mc, nx, term = dqn.read_memory(0, 10)
    np.testing.assert_array_equal(s, st[:10]); np.testing.assert_array_equal(a, ac)
    np.testing.assert_array_equal(mc, exp)
    np.testing.assert_array_equal(term, [0] * 9 + [1])
    np.testing.assert_array_equal(nx[:9], st[1:10]); assert not nx[9].any()
    with pytest.raises(pkg.DQNFatal):
        dqn.LabelTransitions([])                              # CHECK_GT(transitions.size(), 0)
    dqn.close()


def test_select_actions_any_batch_and_epsilon(pkg, gpu):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64))
    for n in (1, 5, 32, 33, 100, 257):
        s = synth_states(rng, n, 59)
        np.testing.assert_allclose(dqn.SelectActionGreedily(s), orc.actor_forward(s), atol=1e-4)
        np.testing.assert_allclose(dqn.SelectActionGreedily(s, net=pkg.ACTOR_TARGET), orc.actor_forward(s, net=2), atol=1e-4)
    s = synth_states(rng, 8, 59)
    np.testing.assert_allclose(dqn.SelectActions(s, 0.0), orc.actor_forward(s), atol=1e-4)   # epsilon 0: greedy
    rnd = dqn.SelectActions(s, 1.0)                                                         # epsilon 1: random
    assert rnd.shape == (8, 10) and np.all(np.abs(rnd[:, :4]) <= 1) and np.all(rnd[:, 8] >= 0)
    assert dqn.SelectAction(s[0], 0.0).shape == (10,)
    with pytest.raises(pkg.DQNFatal):
        dqn.SelectActions(s, 1.5)                               # CHECK(epsilon >= 0 && epsilon <= 1)
    # CHECK_LE(states_batch.size(), kMinibatchSize), src/dqn.cpp:699 — kept; select_actions_cap widens it (-1: any batch)
    s33 = synth_states(rng, 33, 59)
    with pytest.raises(pkg.DQNFatal, match="kMinibatchSize"):
        dqn.SelectActions(s33, 0.0)
    assert dqn.SelectActions(s33[:32], 0.0).shape == (32, 10)
    dqn.select_actions_cap = -1
    np.testing.assert_allclose(dqn.SelectActions(s33, 0.0), orc.actor_forward(s33), atol=1e-4)
    dqn.select_actions_cap = 40
    assert dqn.SelectActions(s33, 1.0).shape == (33, 10)
    with pytest.raises(pkg.DQNFatal):
        dqn.SelectActions(synth_states(rng, 41, 59), 0.0)
    dqn.close(); orc.close()


def test_error_convention(pkg, gpu):
    dqn = pkg.DQN(59, minibatch=32, hidden=(64,), memory=100)
    with pytest.raises(pkg.DQNFatal, match="empty"):
        dqn.UpdateActorCritic()                                 # nothing to sample
    rng = np.random.default_rng(2)
    dqn.add_transitions_arrays(*_mk(rng, 50, 59))
    with pytest.raises(pkg.DQNFatal, match="out of range"):
        dqn.UpdateActorCritic(np.full(32, 50))                  # index == size
    with pytest.raises(pkg.DQNFatal):
        dqn.add_transitions_arrays(*_mk(rng, 100, 59))          # batch >= capacity: deque would underflow
    with pytest.raises(pkg.DQNFatal):
        dqn.set_params(0, np.zeros(3, np.float32))              # wrong parameter count
    assert dqn.Update() is None                                 # below memory_threshold (src/dqn.cpp:800)
    with pytest.raises(pkg.DQNFatal):
        pkg.DQN(59, minibatch=48)                               # not a multiple of 32
    with pytest.raises(pkg.DQNFatal):
        pkg.DQN(59, hidden=(100,))                              # not a multiple of 64
    dqn.close()


def test_params_roundtrip_and_clone(pkg, gpu):
    dqn = pkg.DQN(68, minibatch=32, hidden=(128, 64), memory=100)
    rng = np.random.default_rng(3)
    for net in range(4):
        for kind in ((0, 1, 2) if net < 2 else (0,)):
            w = rng.standard_normal(dqn.param_count(net)).astype(np.float32)
            dqn.set_params(net, w, kind)
            np.testing.assert_array_equal(dqn.get_params(net, kind), w)
    assert dqn.param_count(0) == 128 * 68 + 128 + 64 * 128 + 64 + 10 * 64 + 10
    assert dqn.param_count(1) == 128 * 78 + 128 + 64 * 128 + 64 + 64 + 1
    dqn.CloneNet(0); dqn.CloneNet(1)
    np.testing.assert_array_equal(dqn.get_params(2), dqn.get_params(0))
    np.testing.assert_array_equal(dqn.get_params(3), dqn.get_params(1))
    dqn.set_iters(7, 9); assert (dqn.actor_iter(), dqn.critic_iter(), dqn.max_iter(), dqn.min_iter()) == (7, 9, 9, 7)
    dqn.close()


def test_graph_replay_equals_eager(pkg, gpu):
    """The captured hipGraph replays the identical kernel sequence: bit-identical parameters."""
    res = []
    for use_graph in (False, True):
        dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64), seed=9, use_graph=use_graph)
        for it in range(5):
            dqn.UpdateActorCritic(rng.integers(0, 2048, size=32))
        for it in range(5):
            dqn.UpdateActorCritic()                                # device-side Philox sampling
        res.append([dqn.get_params(n).copy() for n in range(4)] + [np.array(dqn.read_stats())])
        dqn.close(); orc.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_random_sequence_of_ring_operations(pkg, gpu):
    """200 random AddTransitions / AddTransition / ClearReplayMemory calls against the oracle's deque model:
    size and full contents (logical order) after every call."""
    S, cap = 58, 257
    rng = np.random.default_rng(99)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=cap)
    orc = c_oracle.Oracle(B=32, S=S, hidden=(64,), capacity=cap)
    for step in range(200):
        op = rng.random()
        if op < 0.6:
            n = int(rng.integers(1, cap))                     # up to cap-1 transitions at once
            d = _mk(rng, n, S)
            dqn.add_transitions_arrays(*d); orc.add_transitions(*d)
        elif op < 0.95:
            d = _mk(rng, 1, S)
            t = pkg.Transition(d[0][0], d[1][0], float(d[2][0]), float(d[3][0]), None if d[5][0] else d[4][0])
            dqn.AddTransition(t)
            orc.add_transition(d[0][0], d[1][0], d[2][0], d[3][0], d[4][0], d[5][0])
        else:
            dqn.ClearReplayMemory(); orc.clear_memory()
        n = orc.memory_size()
        assert dqn.memory_size() == n <= cap
        if n and step % 5 == 0:
            for x, y in zip(dqn.read_memory(0, n), orc.read_memory(0, n)):
                np.testing.assert_array_equal(x, y)
    with pytest.raises(pkg.DQNFatal):
        dqn.add_transitions_arrays(*_mk(rng, cap, S))          # a batch of `capacity` can never fit (src/dqn.cpp:776)
    dqn.close(); orc.close()
