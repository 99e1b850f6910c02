"""The two restatements against each other: hand-written fp32 backward (C) vs float64 autograd
(PyTorch), on random nets and transitions.  CPU."""
import numpy as np
import pytest

from oracle import c_oracle, torch_ref
from synth import synth_replay


@pytest.mark.parametrize("B,S,hidden,ws", [(32, 59, (64, 64, 64, 64), 10.0), (32, 68, (128, 64), 8.0),
                                            (64, 77, (64,), 10.0), (32, 58, (256, 128, 64), 6.0)])
def test_c_oracle_vs_float64_autograd(B, S, hidden, ws):
    rng = np.random.default_rng(B + S + len(hidden))
    o = c_oracle.Oracle(B=B, S=S, hidden=hidden, capacity=2048)
    t = torch_ref.TorchRef(B=B, S=S, hidden=hidden)
    for net, actor in ((0, True), (1, False)):
        w = torch_ref.init_params_np(rng, S, hidden, actor) * ws
        o.set_params(net, w); o.clone_to_target(net); t.set_params(net, w); t.set_params(net + 2, w)
    s, a, r, mc, nx, term = synth_replay(rng, 1024, S, mean_len=10)
    o.add_transitions(s, a, r, mc, nx, term)
    for it in range(4):
        idx = rng.integers(0, 1024, size=B)
        l1, q1 = o.update(idx)
        l2, q2 = t.update(s[idx], a[idx], r[idx], mc[idx], nx[idx], term[idx])
        assert abs(l1 - l2) <= 2e-6 * max(1.0, abs(l2)) and abs(q1 - q2) <= 1e-5
        for k in ("q_target", "y", "q_train", "q_policy", "actor_out", "dq_da"):
            np.testing.assert_allclose(o.debug_read(k).ravel(), t.dbg[k].numpy().ravel(), rtol=2e-4, atol=2e-5, err_msg=k)
    for net in range(4):
        np.testing.assert_allclose(o.get_params(net), t.get_params(net), rtol=1e-4, atol=2e-6)
    assert o.get_iters() == (4, 4) and t.iter == [4, 4]
    o.close()


def test_soft_update_frequency():
    # src/dqn.cpp:967: targets move only when max_iter() % soft_update_freq == 0
    B, S, hidden = 32, 59, (64,)
    rng = np.random.default_rng(0)
    o = c_oracle.Oracle(B=B, S=S, hidden=hidden, capacity=512, soft_update_freq=2)
    for net, actor in ((0, True), (1, False)):
        o.set_params(net, torch_ref.init_params_np(rng, S, hidden, actor) * 10); o.clone_to_target(net)
    o.add_transitions(*synth_replay(rng, 256, S, mean_len=10))
    t0 = o.get_params(3).copy()
    o.update(rng.integers(0, 256, size=B))          # iter 1: no soft update
    np.testing.assert_array_equal(o.get_params(3), t0)
    o.update(rng.integers(0, 256, size=B))          # iter 2: soft update
    assert np.abs(o.get_params(3) - t0).max() > 0
    o.close()
