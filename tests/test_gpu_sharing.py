"""Multi-agent sharing: DQN::ShareParameters / ShareReplayMemory
(src/dqn.cpp:1036-1083, called from src/dqn_main.cpp:305-323).

The C oracle has no aliasing; the reference's semantics (Blob::ShareData on the first layers of the
online AND target nets; gradients and Adam history per learner) are emulated with two oracles by
copying the shared prefix of the dense parameter vectors from the learner that just updated to its
teammate — sequential Hogwild."""
import numpy as np
import pytest

from helpers import make_pair
from synth import synth_replay

pytestmark = pytest.mark.gpu

S, HID, B = 59, (128, 64, 64, 64), 32


def _prefix(S_in, hidden, n_layers):
    """dense parameter count of the first n tower layers (weights + biases, Caffe order)."""
    dims = (S_in,) + tuple(hidden)
    return sum(dims[i + 1] * dims[i] + dims[i + 1] for i in range(n_layers))


def _sync(src, dst, na, nc):
    """oracle-side stand-in for the aliasing: dst's shared layers := src's (online and target)."""
    for net, n, sin in ((0, na, S), (1, nc, S + 10), (2, na, S), (3, nc, S + 10)):
        k = _prefix(sin, HID, n)
        p = dst.get_params(net).copy()
        p[:k] = src.get_params(net)[:k]
        dst.set_params(net, p)


@pytest.mark.parametrize("use_graph", [0, 1])
def test_share_parameters_two_learners(pkg, gpu, use_graph):
    na, nc = 2, 1
    A, oA, dA, rngA = make_pair(pkg, B=B, S=S, hidden=HID, seed=11, wscale=5.0, use_graph=use_graph)
    Bl, oB, dB, rngB = make_pair(pkg, B=B, S=S, hidden=HID, seed=12, wscale=5.0, use_graph=use_graph)
    # one update each BEFORE sharing, so that captured graphs (if any) exist and must be rebuilt
    for l, o, r in ((A, oA, rngA), (Bl, oB, rngB)):
        idx = r.integers(0, 2048, size=B)
        l.UpdateActorCritic(idx); o.update(idx)
    A.ShareParameters(Bl, na, nc)
    _sync(oA, oB, na, nc)
    ka, kc = _prefix(S, HID, na), _prefix(S + 10, HID, nc)
    for net, k in ((0, ka), (1, kc), (2, ka), (3, kc)):
        np.testing.assert_array_equal(Bl.get_params(net)[:k], A.get_params(net)[:k])
    before_tail = [Bl.get_params(n)[k:].copy() for n, k in ((0, ka), (1, kc))]
    # A updates: B sees the shared layers move, keeps its own upper layers
    idx = rngA.integers(0, 2048, size=B)
    A.UpdateActorCritic(idx); oA.update(idx); _sync(oA, oB, na, nc)
    for (n, k), t in zip(((0, ka), (1, kc)), before_tail):
        np.testing.assert_array_equal(Bl.get_params(n)[:k], A.get_params(n)[:k])
        np.testing.assert_array_equal(Bl.get_params(n)[k:], t)
    # alternate updates; each learner's solver writes the shared weights with its own Adam history
    n_it = 3
    for it in range(n_it):
        idx = rngB.integers(0, 2048, size=B)
        l1, q1 = Bl.UpdateActorCritic(idx); l2, q2 = oB.update(idx); _sync(oB, oA, na, nc)
        assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)) and abs(q1 - q2) <= 1e-4 + 1e-5 * abs(q2)
        idx = rngA.integers(0, 2048, size=B)
        l1, q1 = A.UpdateActorCritic(idx); l2, q2 = oA.update(idx); _sync(oA, oB, na, nc)
        assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)) and abs(q1 - q2) <= 1e-4 + 1e-5 * abs(q2)
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for l, o in ((A, oA), (Bl, oB)):
        for net in range(4):
            d = np.abs(l.get_params(net) - o.get_params(net))
            assert d.max() <= 2 * (n_it + 2) * lr[net] + 1e-6, (net, d.max())
            assert d.mean() <= 0.02 * lr[net] + 1e-8, (net, d.mean())
        for kind in (1, 2):                      # Adam history stays per learner
            for net in (0, 1):
                a, b = l.get_params(net, kind), o.get_params(net, kind)
                np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * np.abs(b).max())
    assert np.abs(A.get_params(0, 1)[:ka] - Bl.get_params(0, 1)[:ka]).max() > 0    # m differs: ShareDiff is commented out (:1044)
    # acting through the shared layers
    s = dA[0][:40]
    np.testing.assert_allclose(Bl.SelectActionGreedily(s), oB.actor_forward(s), atol=1e-4)
    # the owner cannot go away first
    with pytest.raises(pkg.DQNFatal):
        A.close()
    # un-share: B keeps running on its own (stale) copy of those layers
    A.ShareParameters(Bl, 0, 0)
    w = A.get_params(0).copy()
    Bl.UpdateActorCritic(rngB.integers(0, 2048, size=B))
    np.testing.assert_array_equal(A.get_params(0), w)
    Bl.close(); A.close(); oA.close(); oB.close()


def test_share_parameters_rejects(pkg, gpu):
    A = pkg.DQN(S, minibatch=B, hidden=HID, memory=1000)
    C2 = pkg.DQN(S, minibatch=B, hidden=(128, 64), memory=1000)
    D = pkg.DQN(S, minibatch=B, hidden=HID, memory=1000)
    with pytest.raises(pkg.DQNFatal):
        A.ShareParameters(C2, 1, 1)                      # different shapes
    with pytest.raises(pkg.DQNFatal):
        A.ShareParameters(D, 7, 0)                       # CHECK_LT(i, actor_layers.size()), src/dqn.cpp:1060
    with pytest.raises(pkg.DQNFatal):
        A.ShareParameters(D, 5, 0)                       # action_layer without actionpara_layer
    A.ShareParameters(D, 6, 5)                           # everything
    np.testing.assert_array_equal(A.get_params(0), D.get_params(0))
    np.testing.assert_array_equal(A.get_params(1), D.get_params(1))
    D.close(); C2.close(); A.close()


def test_share_replay_memory(pkg, gpu):
    """other.replay_memory_ = replay_memory_ (a shared_ptr, src/dqn.hpp:187): one deque, two users."""
    rng = np.random.default_rng(3)
    A, oA, dA, _ = make_pair(pkg, B=B, S=S, hidden=HID, seed=21, n_replay=500, capacity=1000)
    Bl, oB, dB, _ = make_pair(pkg, B=B, S=S, hidden=HID, seed=22, n_replay=300, capacity=2000)
    assert Bl.memory_size() == 300
    A.ShareReplayMemory(Bl)
    assert Bl.memory_size() == A.memory_size() == 500          # B's own 300 are gone
    for x, y in zip(Bl.read_memory(0, 500), A.read_memory(0, 500)):
        np.testing.assert_array_equal(x, y)
    d = synth_replay(rng, 200, S, mean_len=7)
    Bl.add_transitions_arrays(*d); oA.add_transitions(*d)      # B writes, A sees it
    assert A.memory_size() == Bl.memory_size() == 700
    d = synth_replay(rng, 450, S, mean_len=7)
    A.add_transitions_arrays(*d); oA.add_transitions(*d)       # wraps at A's capacity (1000), evicting
    assert A.memory_size() == Bl.memory_size() == oA.memory_size() == 999
    for x, y in zip(Bl.read_memory(0, 999), oA.read_memory(0, 999)):
        np.testing.assert_array_equal(x, y)
    # B samples from the shared memory: same update as an oracle holding B's weights + A's memory
    oB.clear_memory()
    oB.add_transitions(*oA.read_memory(0, 999))
    for it in range(2):
        idx = rng.integers(0, 999, size=B)
        l1, q1 = Bl.UpdateActorCritic(idx); l2, q2 = oB.update(idx)
        assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)) and abs(q1 - q2) <= 1e-4 + 1e-5 * abs(q2)
        for name in ("y", "q_train"):
            np.testing.assert_allclose(Bl.debug_read(name), oB.debug_read(name), rtol=1e-5, atol=1e-4)
    l, q = Bl.UpdateActorCritic()                              # device-side sampling over the shared size
    assert np.isfinite(l) and np.isfinite(q)
    assert Bl.debug_read("idx").max() < 999
    Bl.ClearReplayMemory()
    assert A.memory_size() == 0
    with pytest.raises(pkg.DQNFatal):
        A.close()
    Bl.close(); A.close(); oA.close(); oB.close()


def test_two_agents_share_ring_and_layers_under_load(pkg, gpu):
    """Both agents run their own batched workers into ONE replay memory and update alternately on their
    own streams (the reference's two KeepPlayingGames threads, src/dqn_main.cpp:305-323): the event
    ordering of the shared ring and the aliased layers hold up, nothing faults, everything stays finite."""
    A = pkg.DQN(58, minibatch=64, hidden=(128, 64, 64), memory=5000, seed=1)
    Bl = pkg.DQN(58, minibatch=64, hidden=(128, 64, 64), memory=700, seed=2)
    A.ShareParameters(Bl, 2, 1)
    A.ShareReplayMemory(Bl)
    envA = pkg.EnvFrontEnd(A, 32, max_steps=50, p_end=0.04, seed=3)
    envB = pkg.EnvFrontEnd(Bl, 32, max_steps=50, p_end=0.04, seed=4)
    for it in range(30):
        envA.step(0.5, 10); envB.step(0.5, 10)
        if A.memory_size() >= 500:
            la, qa = A.UpdateActorCritic(); lb, qb = Bl.UpdateActorCritic()
            assert np.isfinite([la, qa, lb, qb]).all()
    sa, sb = envA.stats(), envB.stats()
    assert sa[0] == sb[0] == 30 * 10 * 32
    assert A.memory_size() == Bl.memory_size() == 4999            # 19200 transitions through the owner's 5000 slots
    ka = 58 * 128 + 128 + 128 * 64 + 64
    np.testing.assert_array_equal(A.get_params(0)[:ka], Bl.get_params(0)[:ka])
    assert np.abs(A.get_params(0)[ka:] - Bl.get_params(0)[ka:]).max() > 0
    s = A.read_memory(0, 4999)[0]
    assert np.isfinite(s).all() and (np.abs(s) <= 1 + 1e-6).all()
    envB.close(); envA.close(); Bl.close(); A.close()


def test_share_replay_memory_drops_the_captured_dp_graph(pkg, gpu):
    """the captured data-parallel update bakes in the Ring struct (k_gather takes it by value); after
    ShareReplayMemory a replay of the old graph would keep sampling the sharer's former private ring.  Every path that
    invalidates the update graphs must invalidate that one too."""
    A, oA, dA, _ = make_pair(pkg, B=B, S=S, hidden=HID, seed=21, n_replay=500, capacity=1000)
    U, oU, dU, _ = make_pair(pkg, B=B, S=S, hidden=HID, seed=22, n_replay=300, capacity=2000, use_graph=True)
    E, oE, _, _ = make_pair(pkg, B=B, S=S, hidden=HID, seed=22, n_replay=300, capacity=2000)       # U's eager twin
    for d in (U, E):
        d.dp_init(pkg.DQN.dp_unique_id())
        d.dp_update(None)
    assert U.dp_graph_active() and U.read_stats() == E.read_stats()
    A.ShareReplayMemory(U); A.ShareReplayMemory(E)
    assert not U.dp_graph_active()                       # dropped with the other captured launches
    for it in range(2):
        U.dp_update(None); E.dp_update(None)
        assert U.read_stats() == E.read_stats()          # both sample A's 500 transitions now
        np.testing.assert_array_equal(U.debug_read("idx"), E.debug_read("idx"))
        np.testing.assert_array_equal(U.debug_read("y"), E.debug_read("y"))
    assert U.dp_graph_active()
    for net in range(4):
        np.testing.assert_array_equal(U.get_params(net), E.get_params(net))
    for x in (U, E, A, oA, oU, oE):
        x.close()
