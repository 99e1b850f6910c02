"""Batched env front-end (include/dqnhip_env.h) vs its CPU restatement on the same synthetic
state stream: per-step actions/rewards, and the replay contents after episodes were labelled
and appended (LabelTransitions + AddTransitions in worker order)."""
import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workers,eps,S,hidden,wscale", [
    (5, 0.0, 59, (128, 64, 64, 64), 10.0), (64, 0.3, 59, (128, 64, 64, 64), 10.0), (100, 1.0, 59, (128, 64, 64, 64), 10.0),
    (64, 0.3, 68, (1024, 1024, 1024, 1024), 3.0),      # BASELINE.json configs[2]: 1v1 (S = 68), 64 workers, the 4x1024 tower
    (64, 0.1, 77, (1024, 1024, 1024, 1024), 3.0),      # configs[3]'s state size (2v1, S = 77)
])
def test_env_front_end_matches_oracle(pkg, gpu, workers, eps, S, hidden, wscale):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=S, hidden=hidden, n_replay=100, capacity=20000, wscale=wscale)
    kw = dict(max_steps=40, unum=7, p_end=0.05, p_goal=0.4, seed=11)
    env = pkg.EnvFrontEnd(dqn, workers, **kw)
    oenv = c_oracle.OracleEnv(orc, workers, **kw)
    np.testing.assert_allclose(env.debug_read("state"), oenv.read()["state"], atol=1e-6)
    for step in range(60):
        env.step(eps); oenv.step(eps)
        o = oenv.read()
        np.testing.assert_array_equal(env.debug_read("action").astype(np.int32), o["action"])       # indices exact
        np.testing.assert_allclose(env.debug_read("arg1"), o["arg1"], atol=1e-4)
        np.testing.assert_allclose(env.debug_read("reward"), o["reward"], atol=2e-5)
        np.testing.assert_array_equal(env.debug_read("episode_len").astype(np.int32), o["episode_len"])
        np.testing.assert_allclose(env.debug_read("state"), o["state"], atol=1e-6)
    s1, s2 = env.stats(), oenv.stats()
    assert s1[0] == s2[0] == 60 * workers and s1[1] == s2[1] and s1[3] == s2[3] and s1[1] > 0
    assert abs(s1[2] - s2[2]) < 1e-2
    assert dqn.memory_size() == orc.memory_size()
    a, b = dqn.read_memory(0, dqn.memory_size()), orc.read_memory(0, orc.memory_size())
    np.testing.assert_allclose(a[0], b[0], atol=1e-6)          # states
    np.testing.assert_allclose(a[1], b[1], atol=1e-4)          # actor outputs
    np.testing.assert_allclose(a[2], b[2], atol=2e-5)          # rewards
    np.testing.assert_allclose(a[3], b[3], atol=2e-4)          # Monte-Carlo labels
    np.testing.assert_allclose(a[4], b[4], atol=1e-6)          # next states
    np.testing.assert_array_equal(a[5], b[5])                  # terminal flags
    # and the learner keeps working on what the workers produced
    loss, q = dqn.UpdateActorCritic()
    assert np.isfinite(loss) and np.isfinite(q)
    env.close(); oenv.close(); dqn.close(); orc.close()


@pytest.mark.parametrize("workers,use_graph", [(1024, False), (2048, False), (2048, True)])
def test_env_many_workers_match_oracle(pkg, gpu, workers, use_graph):
    """BASELINE.json configs[4]'s worker count (2048; 1024 beside it): above 512 workers the step takes the tiled head
    kernel, the flush in its own launch and a separate commit.  Transition by transition against the oracle for a
    few batched steps with a small tower (episodes of <= 10 steps, so every worker finishes several)."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=58, hidden=(128, 64, 64, 64), n_replay=100, capacity=120000, wscale=10.0,
                                    use_graph=use_graph)
    kw = dict(max_steps=10, unum=7, p_end=0.1, p_goal=0.4, seed=13)
    env = pkg.EnvFrontEnd(dqn, workers, **kw)
    oenv = c_oracle.OracleEnv(orc, workers, **kw)
    total = 0
    for eps, n in ((0.2, 1), (0.2, 5), (0.0, 17), (1.0, 2)):
        env.step(eps, n); oenv.step(eps, n); total += n
        o = oenv.read()
        np.testing.assert_array_equal(env.debug_read("action").astype(np.int32), o["action"])       # indices exact
        np.testing.assert_allclose(env.debug_read("arg1"), o["arg1"], atol=1e-4, rtol=1e-5)      # parameters reach +-180: fp32 round-off of a different summation order
        np.testing.assert_allclose(env.debug_read("reward"), o["reward"], atol=5e-5)      # differences of distances ~100: the max over 2048 workers is a few 1e-5
        np.testing.assert_array_equal(env.debug_read("episode_len").astype(np.int32), o["episode_len"])
        np.testing.assert_allclose(env.debug_read("state"), o["state"], atol=1e-6)
        assert dqn.memory_size() == orc.memory_size()
    s1, s2 = env.stats(), oenv.stats()
    assert s1[0] == s2[0] == total * workers and s1[1] == s2[1] > 2 * workers and s1[3] == s2[3]
    n = dqn.memory_size()
    a, b = dqn.read_memory(0, n), orc.read_memory(0, n)
    np.testing.assert_allclose(a[0], b[0], atol=1e-6); np.testing.assert_allclose(a[1], b[1], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(a[2], b[2], atol=5e-5); np.testing.assert_allclose(a[3], b[3], atol=4e-4)
    np.testing.assert_allclose(a[4], b[4], atol=1e-6); np.testing.assert_array_equal(a[5], b[5])
    env.close(); oenv.close(); dqn.close(); orc.close()


def test_env_graph_replay_matches_oracle(pkg, gpu):
    """use_graph: the batched step is replayed as captured hipGraphs (16-step and 1-step), epsilon is
    a device scalar; 37 = 2 x 16 + 5 exercises both graphs, a changed epsilon the scalar."""
    workers = 64
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64), n_replay=100, capacity=20000,
                                    use_graph=True)
    kw = dict(max_steps=40, unum=7, p_end=0.05, p_goal=0.4, seed=11)
    env = pkg.EnvFrontEnd(dqn, workers, **kw)
    oenv = c_oracle.OracleEnv(orc, workers, **kw)
    total = 0
    for eps, n in ((0.3, 37), (0.05, 16), (1.0, 3)):
        env.step(eps, n); oenv.step(eps, n); total += n
        o = oenv.read()
        np.testing.assert_array_equal(env.debug_read("action").astype(np.int32), o["action"])
        np.testing.assert_array_equal(env.debug_read("episode_len").astype(np.int32), o["episode_len"])
        np.testing.assert_allclose(env.debug_read("state"), o["state"], atol=1e-6)
    s1, s2 = env.stats(), oenv.stats()
    assert s1[0] == s2[0] == total * workers and s1[1] == s2[1] and s1[3] == s2[3] and s1[1] > 0
    assert dqn.memory_size() == orc.memory_size()
    a, b = dqn.read_memory(0, dqn.memory_size()), orc.read_memory(0, orc.memory_size())
    np.testing.assert_allclose(a[0], b[0], atol=1e-6); np.testing.assert_allclose(a[3], b[3], atol=2e-4)
    np.testing.assert_array_equal(a[5], b[5])
    loss, q = dqn.UpdateActorCritic()                       # the captured update and the captured steps coexist
    env.step(0.1, 20); oenv.step(0.1, 20)
    assert dqn.memory_size() == orc.memory_size() and np.isfinite(loss)
    env.close(); oenv.close(); dqn.close(); orc.close()


@pytest.mark.parametrize("workers,use_graph,S,hidden", [
    (48, False, 59, (128, 64, 64, 64)), (48, True, 59, (128, 64, 64, 64)), (600, False, 59, (128, 64, 64, 64)),
    # wide towers: k_env_step(t) computes the first layer of step t+1 itself and the flush rides on the SECOND layer's
    # launch (S = 59: one 64-wide k chunk; S = 68: two)
    (48, False, 59, (256, 512, 128)), (64, True, 68, (512, 512, 512, 64)), (33, True, 59, (64, 512)),
])
def test_env_step_sequences_match_single_steps(pkg, gpu, workers, use_graph, S, hidden):
    """Inside a sequence of steps the episode flush of step t rides in a tower-layer launch of step t+1 (<= 512 workers;
    above that the flush keeps its own launch and the dedicated head kernel runs): sequences of 1, 2, 7 and 19 steps —
    eager and graph-replayed — must leave the same workers and the same replay as the oracle stepping one at a time."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=S, hidden=hidden, n_replay=100, capacity=60000, use_graph=use_graph)
    kw = dict(max_steps=30, unum=7, p_end=0.08, p_goal=0.4, seed=5)
    env = pkg.EnvFrontEnd(dqn, workers, **kw)
    oenv = c_oracle.OracleEnv(orc, workers, **kw)
    for n in (1, 2, 7, 19, 1, 16, 3):
        env.step(0.2, n)
        for _ in range(n):
            oenv.step(0.2)
        o = oenv.read()
        np.testing.assert_array_equal(env.debug_read("action").astype(np.int32), o["action"])
        np.testing.assert_array_equal(env.debug_read("episode_len").astype(np.int32), o["episode_len"])
        np.testing.assert_allclose(env.debug_read("state"), o["state"], atol=1e-6)
        assert dqn.memory_size() == orc.memory_size()
    a, b = dqn.read_memory(0, dqn.memory_size()), orc.read_memory(0, orc.memory_size())
    np.testing.assert_allclose(a[0], b[0], atol=1e-6); np.testing.assert_allclose(a[2], b[2], atol=2e-5)
    np.testing.assert_allclose(a[3], b[3], atol=2e-4); np.testing.assert_allclose(a[4], b[4], atol=1e-6)
    np.testing.assert_array_equal(a[5], b[5])
    assert env.stats()[1] == oenv.stats()[1] > 0
    env.close(); oenv.close(); dqn.close(); orc.close()


def test_env_wraps_ring_and_rejects_bad_config(pkg, gpu):
    dqn = pkg.DQN(59, minibatch=32, hidden=(64,), memory=3000)
    with pytest.raises(pkg.DQNFatal):
        pkg.EnvFrontEnd(dqn, 64, max_steps=100)                # workers*max_steps >= capacity
    env = pkg.EnvFrontEnd(dqn, 16, max_steps=50, p_end=0.1)
    env.step(0.5, 400)                                         # 6400 transitions through a 3000-slot ring
    steps, episodes, _, _ = env.stats()
    assert steps == 6400 and episodes > 50
    assert 2900 - 16 * 50 <= dqn.memory_size() <= 2999          # AddTransitions keeps <= cap-1
    with pytest.raises(pkg.DQNFatal):
        env.step(1.5)
    dqn2 = pkg.DQN(50, minibatch=32, hidden=(64,), memory=3000)
    with pytest.raises(pkg.DQNFatal, match="56"):
        pkg.EnvFrontEnd(dqn2, 4)                               # HFOGameState needs state indices up to 55
    env.close(); dqn.close(); dqn2.close()
