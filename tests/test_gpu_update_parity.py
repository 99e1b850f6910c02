"""HIP path vs CPU oracle on identical weights, transitions and sampled indices
(SURVEY.md F5).  Tolerances: Q-values 1e-4 abs (BASELINE.json north_star), action
indices exact."""
import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle

pytestmark = pytest.mark.gpu

QTOL = 1e-4   # north_star: "Q-values within 1e-4 fp32" (absolute, for |Q| = O(1))
QRTOL = 1e-5  # plus 1e-5 relative so the bound stays meaningful when the random nets emit |Q| >> 1


def _check_update(dqn, orc, idx):
    l1, q1 = dqn.UpdateActorCritic(idx)
    l2, q2 = orc.update(idx)
    for name in ("q_target", "y", "q_train", "q_policy"):
        np.testing.assert_allclose(dqn.debug_read(name), orc.debug_read(name), rtol=QRTOL, atol=QTOL, err_msg=name)
    np.testing.assert_array_equal(dqn.debug_read("idx").astype(np.int64), np.asarray(idx, np.int64))
    np.testing.assert_array_equal(dqn.debug_read("terminal"), orc.debug_read("terminal"))
    a1, a2 = dqn.debug_read("actor_out"), orc.debug_read("actor_out")
    np.testing.assert_allclose(a1, a2, rtol=1e-4, atol=1e-4)
    g1, g2 = dqn.debug_read("dq_da"), orc.debug_read("dq_da")
    np.testing.assert_allclose(g1, g2, rtol=2e-3, atol=1e-6)
    assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)), (l1, l2)
    assert abs(q1 - q2) <= QTOL + QRTOL * abs(q2), (q1, q2)


@pytest.mark.parametrize("shape", [
    dict(B=32, S=59, hidden=(1024, 512, 256, 128)),        # reference defaults (src/dqn.hpp:19, dqn.cpp:425)
    dict(B=32, S=68, hidden=(128, 64, 64, 64)),            # 1v1 state size, small tower
    dict(B=64, S=77, hidden=(256, 128), ),                 # 2v1 state size, 2-layer tower
    # BASELINE.json config #2.  wscale 2 (weights N(0, 0.02^2)): at 5x the 4x1024 critic's loss
    # explodes to 5e4 after one lr=1e-3 Adam step and HIP, the C oracle and a float64 reference
    # then differ from each other by ReLU-mask flips in different rows (all three measured).
    dict(B=256, S=58, hidden=(1024, 1024, 1024, 1024), wscale=2.0),
])
def test_update_matches_oracle(pkg, gpu, shape):
    shape = dict(shape)
    shape.setdefault("wscale", 5.0)
    dqn, orc, data, rng = make_pair(pkg, n_replay=2048, **shape)
    B = shape["B"]
    for it in range(4):
        idx = rng.integers(0, 2048, size=B)
        _check_update(dqn, orc, idx)
    # Adam's normalised step m/(sqrt(v)+eps) is O(1) whatever |g| is, so an element whose gradient
    # is at fp32-roundoff level may legitimately move differently by up to lr per update; everything
    # else must agree to 1e-6.
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net in range(4):
        d = np.abs(dqn.get_params(net) - orc.get_params(net))
        assert d.max() <= 4 * lr[net] + 1e-6, (net, d.max())
        assert (d > 1e-6).mean() <= 1e-4, (net, (d > 1e-6).mean())
    for kind in (1, 2):   # Adam m, v
        for net in (0, 1):
            a, b = dqn.get_params(net, kind), orc.get_params(net, kind)
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * np.abs(b).max())
    assert dqn.actor_iter() == 4 and dqn.critic_iter() == 4
    dqn.close(); orc.close()


def test_action_indices_bit_exact(pkg, gpu):
    """GetAction argmax over {DASH,TURN,KICK} on 512 probe states: indices identical."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(1024, 512, 256, 128), wscale=10.0)
    probe = data[0][:512]
    out_h = dqn.SelectActionGreedily(probe)
    out_o = orc.actor_forward(probe)
    assert np.abs(out_h - out_o).max() <= QTOL
    act_o, a1_o, a2_o = c_oracle.get_action(out_o)
    acts_h = [pkg.GetAction(o) for o in out_h]
    assert [a.action for a in acts_h] == list(act_o)
    assert 2 not in act_o                      # TACKLE is never returned (src/dqn.cpp:198)
    # margin between best and second-best logit, so the equality above is not luck
    lg = out_o[:, [0, 1, 3]]
    srt = np.sort(lg, axis=1)
    assert (srt[:, -1] - srt[:, -2]).min() > 10 * np.abs(out_h - out_o).max()
    dqn.close(); orc.close()


def test_deterministic_replay(pkg, gpu):
    """Same weights + same indices twice -> bit-identical parameters (no atomics, fixed
    reduction trees)."""
    res = []
    for rep in range(2):
        dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64), seed=5)
        for it in range(3):
            dqn.UpdateActorCritic(rng.integers(0, 2048, size=32))
        res.append([dqn.get_params(n).copy() for n in range(4)])
        dqn.close(); orc.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_critic_forward_and_evaluate(pkg, gpu):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64))
    s, a = data[0][:100], data[1][:100]
    np.testing.assert_allclose(dqn.CriticForward(s, a), orc.critic_forward(s, a), atol=QTOL)
    assert abs(dqn.EvaluateAction(s[0], a[0]) - orc.critic_forward(s[:1], a[:1])[0]) <= QTOL
    dqn.close(); orc.close()
