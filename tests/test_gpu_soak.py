"""Workers feeding the learner for many steps: ring wrap-around under the env front-end, captured
update + captured env steps interleaved, both precisions (the fp16 learner pads its input panels to
128 columns while the replay ring keeps 64-wide rows)."""
import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_env_and_updates_soak(pkg, gpu, precision, use_graph):
    dqn = pkg.DQN(58, minibatch=128, hidden=(256, 128), memory=6000, seed=3, use_graph=use_graph, precision=precision)
    env = pkg.EnvFrontEnd(dqn, 64, max_steps=60, p_end=0.03, seed=9)
    n_upd = 0
    for it in range(40):
        env.step(max(0.1, 1.0 - it / 20.0), 25)
        if dqn.memory_size() >= 1000:
            for _ in range(5):
                dqn.update_async(None); n_upd += 1
            loss, q = dqn.read_stats()
            assert np.isfinite(loss) and np.isfinite(q), (it, loss, q)
    steps, eps, rsum, goals = env.stats()
    assert steps == 40 * 25 * 64 and eps > 100
    assert dqn.memory_size() == 5999                       # wrapped many times (64000 transitions through 6000 slots)
    assert dqn.actor_iter() == n_upd > 100
    s, a, r, mc, nx, term = dqn.read_memory(0, 5999)
    assert np.isfinite(s).all() and np.isfinite(mc).all() and (np.abs(s) <= 1.0 + 1e-6).all()
    assert term.sum() > 50 and not nx[term.astype(bool)].any()
    for net in range(4):
        assert np.isfinite(dqn.get_params(net)).all()
    env.close(); dqn.close()


def test_env_on_fp16_learner_matches_oracle(pkg, gpu):
    """Acting stays on the fp32 master weights with the exact-fp32 kernels in fp16 mode, so the
    front-end of an fp16 learner reproduces the oracle's transitions like the fp32 one does."""
    dqn, orc, data, rng = make_pair(pkg, B=128, S=59, hidden=(128, 128), n_replay=100, capacity=20000, precision="fp16")
    kw = dict(max_steps=40, unum=7, p_end=0.05, p_goal=0.4, seed=11)
    env = pkg.EnvFrontEnd(dqn, 48, **kw)
    oenv = c_oracle.OracleEnv(orc, 48, **kw)
    for step in range(50):
        env.step(0.3); oenv.step(0.3)
    o = oenv.read()
    np.testing.assert_array_equal(env.debug_read("action").astype(np.int32), o["action"])
    assert dqn.memory_size() == orc.memory_size()
    a, b = dqn.read_memory(0, dqn.memory_size()), orc.read_memory(0, orc.memory_size())
    np.testing.assert_allclose(a[0], b[0], atol=1e-6); np.testing.assert_allclose(a[4], b[4], atol=1e-6)
    np.testing.assert_allclose(a[3], b[3], atol=2e-4); np.testing.assert_array_equal(a[5], b[5])
    env.close(); oenv.close(); dqn.close(); orc.close()
