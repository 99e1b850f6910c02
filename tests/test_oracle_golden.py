"""C oracle (fp32 restatement) vs the committed golden vectors (float64 autograd restatement,
tests/golden/make_golden.py).  Runs on CPU.  PARITY UNPINNED against the reference itself — see
oracle/dqn_oracle.c."""
import glob
import os

import numpy as np
import pytest

from oracle import c_oracle
from synth import det_indices, det_params, det_replay, det_uniform

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_case(path):
    g = np.load(path)
    B, S, hidden, ws, seed = int(g["B"]), int(g["S"]), tuple(int(x) for x in g["hidden"]), float(g["wscale"]), int(g["seed"])
    n_rep, n_upd = int(g["n_replay"]), int(g["n_updates"])
    w = {net: det_params(seed + net, S, hidden, net == 0, ws) for net in (0, 1)}
    data = det_replay(seed, n_rep, S)
    idxs = det_indices(seed, n_upd, B, n_rep)
    # the regenerated inputs are the ones the fixture was made from
    assert abs(w[0].astype(np.float64).sum() - float(g["check_w_actor_sum"])) < 1e-9
    assert abs(data[0].astype(np.float64).sum() + data[1].astype(np.float64).sum() - float(g["check_replay_sum"])) < 1e-6
    probe = det_uniform(seed * 10 + 9, 128 * S, -1, 1).reshape(128, S).astype(np.float32)
    return g, dict(B=B, S=S, hidden=hidden), w, data, idxs, probe


def blob_slices(S, hidden, actor):
    from oracle.torch_ref import layout
    off, out = 0, []
    for (n, k) in layout(S if actor else S + 10, hidden, (4, 6) if actor else (1,)):
        out.append((off, off + n * k)); off += n * k
        out.append((off, off + n)); off += n
    return out


def check_against_golden(g, u, learner, read, tol_q=1e-4):
    """learner: anything with get_params(net, kind); read(name) -> minibatch intermediates."""
    for k in ("q_target", "y", "q_train", "q_policy"):
        np.testing.assert_allclose(read(k), g["u%d_%s" % (u, k)], rtol=1e-5, atol=tol_q, err_msg=k)
    np.testing.assert_allclose(read("actor_out"), g["u%d_actor_out" % u], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(read("dq_da"), g["u%d_dq_da" % u], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_c_oracle_matches_golden(path):
    g, shp, w, data, idxs, probe = load_case(path)
    orc = c_oracle.Oracle(capacity=data[2].size + 1, **shp)
    for net in (0, 1):
        orc.set_params(net, w[net]); orc.clone_to_target(net)
    orc.add_transitions(*data)
    for u in range(idxs.shape[0]):
        loss, avgq = orc.update(idxs[u])
        assert abs(loss - float(g["u%d_loss" % u])) <= 1e-5 * max(1.0, abs(loss))
        assert abs(avgq - float(g["u%d_avgq" % u])) <= 1e-5
        check_against_golden(g, u, orc, orc.debug_read)
        for net in range(4):
            v = orc.get_params(net).astype(np.float64)
            norms = np.array([np.linalg.norm(v[a:b]) for a, b in blob_slices(shp["S"], shp["hidden"], net % 2 == 0)])
            np.testing.assert_allclose(norms, g["u%d_w%d_norms" % (u, net)], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(v[g["pick%d" % net]], g["u%d_w%d_pick" % (u, net)], rtol=1e-4, atol=2e-6)
        for net in (0, 1):
            np.testing.assert_allclose(orc.get_params(net, 1)[g["pick%d" % net]], g["u%d_m%d_pick" % (u, net)], rtol=1e-3, atol=1e-8)
            np.testing.assert_allclose(orc.get_params(net, 2)[g["pick%d" % net]], g["u%d_v%d_pick" % (u, net)], rtol=1e-3, atol=1e-12)
    # GetAction on the probe states: indices exact wherever the float64 margin is not roundoff-sized
    ao = orc.actor_forward(probe)
    act, _, _ = c_oracle.get_action(ao)
    np.testing.assert_allclose(ao, g["probe_actor_out"], rtol=1e-4, atol=1e-5)
    safe = g["probe_margin"] > 1e-5
    assert safe.sum() >= 120
    np.testing.assert_array_equal(act[safe], g["probe_action"][safe])
    assert 2 not in act
    orc.close()
