"""HIP path vs the committed golden vectors (float64 autograd restatement) — independent of
the C oracle.  Tolerances: Q-values 1e-4 (north_star), action indices exact."""
import os

import numpy as np
import pytest

from test_oracle_golden import GOLDEN, blob_slices, check_against_golden, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("use_graph", [False, True])
def test_hip_matches_golden(pkg, gpu, path, use_graph):
    g, shp, w, data, idxs, probe = load_case(path)
    dqn = pkg.DQN(shp["S"], minibatch=shp["B"], hidden=shp["hidden"], memory=data[2].size + 1, use_graph=use_graph)
    for net in (0, 1):
        dqn.set_params(net, w[net]); dqn.CloneNet(net)
    dqn.add_transitions_arrays(*data)
    for u in range(idxs.shape[0]):
        loss, avgq = dqn.UpdateActorCritic(idxs[u])
        assert abs(loss - float(g["u%d_loss" % u])) <= 1e-5 * max(1.0, abs(loss))
        assert abs(avgq - float(g["u%d_avgq" % u])) <= 1e-5
        check_against_golden(g, u, dqn, dqn.debug_read)
        for net in range(4):
            v = dqn.get_params(net).astype(np.float64)
            norms = np.array([np.linalg.norm(v[a:b]) for a, b in blob_slices(shp["S"], shp["hidden"], net % 2 == 0)])
            np.testing.assert_allclose(norms, g["u%d_w%d_norms" % (u, net)], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(v[g["pick%d" % net]], g["u%d_w%d_pick" % (u, net)], rtol=1e-4, atol=2e-6)
        for net in (0, 1):
            np.testing.assert_allclose(dqn.get_params(net, 1)[g["pick%d" % net]], g["u%d_m%d_pick" % (u, net)], rtol=1e-3, atol=1e-8)
            np.testing.assert_allclose(dqn.get_params(net, 2)[g["pick%d" % net]], g["u%d_v%d_pick" % (u, net)], rtol=1e-3, atol=1e-12)
    ao = dqn.SelectActionGreedily(probe)
    np.testing.assert_allclose(ao, g["probe_actor_out"], rtol=1e-4, atol=1e-5)
    act = np.array([pkg.GetAction(o).action for o in ao])
    safe = g["probe_margin"] > 1e-5
    np.testing.assert_array_equal(act[safe], g["probe_action"][safe])      # action indices bit-exact
    assert 2 not in act
    dqn.close()
