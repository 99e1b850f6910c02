"""Seeded sweep over shapes and hyper-parameters: every combination runs two full updates on the HIP path
and on the C oracle from identical weights / transitions / indices (the kernels pick different tile
families depending on the dimensions: 16-, 32-, 64-wide tiles, LDS-transposed or direct operands,
narrow first-layer tiles, large-batch head kernels)."""
import numpy as np
import pytest

from helpers import make_pair

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(1, 5))
        hidden = tuple(int(64 * rng.integers(1, 9)) for _ in range(L))
        out.append(dict(B=int(32 * rng.integers(1, 9)), S=int(rng.integers(56, 131)), hidden=hidden,
                        gamma=float(rng.choice([0.9, 0.99, 0.0])), beta=float(rng.choice([0.0, 0.5, 1.0])),
                        tau=float(rng.choice([0.001, 0.05, 1.0])), soft_update_freq=int(rng.choice([1, 2])),
                        clip_grad=float(rng.choice([10.0, 0.05, -1.0])), wscale=float(rng.choice([1.0, 5.0]))))
    return out


@pytest.mark.parametrize("case", _cases(24, 2026), ids=lambda c: "B%d_S%d_%s" % (c["B"], c["S"], "x".join(map(str, c["hidden"]))))
def test_random_configuration_matches_oracle(pkg, gpu, case):
    case = dict(case)
    dqn, orc, data, rng = make_pair(pkg, n_replay=1024, **case)
    B = case["B"]
    for it in range(2):
        idx = rng.integers(0, 1024, size=B)
        l1, q1 = dqn.UpdateActorCritic(idx)
        l2, q2 = orc.update(idx)
        assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)), (it, l1, l2)
        assert abs(q1 - q2) <= 1e-4 + 1e-5 * abs(q2), (it, q1, q2)
        for name in ("q_target", "y", "q_train", "q_policy"):
            np.testing.assert_allclose(dqn.debug_read(name), orc.debug_read(name), rtol=1e-5, atol=1e-4, err_msg=name)
    lr = {0: 1e-5, 1: 1e-3}
    tau = case["tau"]
    for net in range(4):
        d = np.abs(dqn.get_params(net) - orc.get_params(net))
        bound = 2 * lr[net & 1] * (1.0 if net < 2 else max(tau, 1e-3) * 2)
        assert d.max() <= bound + 1e-6, (net, d.max(), bound)
    assert dqn.actor_iter() == 2
    dqn.close(); orc.close()


def _cases16(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(1, 4))
        out.append(dict(B=int(128 * rng.integers(1, 4)), S=int(rng.integers(56, 131)),
                        hidden=tuple(int(128 * rng.integers(1, 5)) for _ in range(L)),
                        gamma=float(rng.choice([0.9, 0.99])), beta=float(rng.choice([0.0, 0.5, 1.0])),
                        clip_grad=float(rng.choice([10.0, -1.0])), wscale=float(rng.choice([2.0, 5.0]))))
    return out


@pytest.mark.parametrize("case", _cases16(10, 77), ids=lambda c: "B%d_S%d_%s" % (c["B"], c["S"], "x".join(map(str, c["hidden"]))))
def test_random_configuration_fp16_tracks_oracle(pkg, gpu, case):
    """The mixed-precision learner on random shapes: finite, and within mixed-precision distance of the
    fp32 oracle (the tight pin of the fp16 pipeline is the emulation test in test_gpu_fp16.py)."""
    case = dict(case)
    dqn, orc, data, rng = make_pair(pkg, n_replay=1024, precision="fp16", **case)
    B = case["B"]
    for it in range(2):
        idx = rng.integers(0, 1024, size=B)
        l1, q1 = dqn.UpdateActorCritic(idx)
        l2, q2 = orc.update(idx)
        assert np.isfinite(l1) and np.isfinite(q1)
        assert abs(l1 - l2) <= 2e-2 * max(1.0, abs(l2)), (it, l1, l2)
        assert abs(q1 - q2) <= 2e-2 * max(1.0, abs(q2)), (it, q1, q2)
        ref = orc.debug_read("q_train")
        assert np.abs(dqn.debug_read("q_train") - ref).max() <= 3e-2 * max(1.0, np.abs(ref).max())
    for net in range(4):
        assert np.isfinite(dqn.get_params(net)).all()
    dqn.close(); orc.close()
