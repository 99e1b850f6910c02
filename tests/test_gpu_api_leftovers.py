"""SURVEY §8 row a17 and the reference's abort semantics on the device path:
SampleStatesFromMemory (src/dqn.cpp:511-523), getActorOutput (:719-732), SampleAction (:180-194),
FilesMatchingRegexp / RemoveSnapshots (src/dqn.hpp:213-224), CHECK(isfinite(target)) (:898) and
CHECK(isfinite(critic_loss)) (:906) surfaced by every entry point."""
import os

import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def test_sample_states_from_memory(pkg, gpu):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(64, 64), n_replay=1500)
    idx = rng.integers(0, 1500, size=200)
    got = dqn.SampleStatesFromMemory(200, idx)
    np.testing.assert_array_equal(got, data[0][idx])                   # explicit indices: the rows themselves
    # device draw: every returned row is a state that is in the memory; successive calls differ
    a, b = dqn.SampleStatesFromMemory(64), dqn.SampleStatesFromMemory(64)
    keys = {row.tobytes() for row in data[0]}
    assert all(row.tobytes() in keys for row in a) and all(row.tobytes() in keys for row in b)
    assert not np.array_equal(a, b)
    with pytest.raises(pkg.DQNFatal, match="out of range"):
        dqn.SampleStatesFromMemory(4, [0, 1, 2, 1500])
    dqn.ClearReplayMemory()
    with pytest.raises(pkg.DQNFatal, match="empty"):
        dqn.SampleStatesFromMemory(4)
    dqn.close(); orc.close()


def test_get_actor_output_reads_the_last_minibatch_forward(pkg, gpu):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64), n_replay=1024)
    idx = rng.integers(0, 1024, size=32)
    mu_before = orc.actor_forward(data[0][idx])                        # mu(s) with the pre-update actor (src/dqn.cpp:910-911)
    mu_t = orc.actor_forward(np.where(data[5][idx][:, None] != 0, 0, data[4][idx]).astype(np.float32), net=c_oracle.ACTOR_TARGET)
    dqn.UpdateActorCritic(idx)
    np.testing.assert_allclose(dqn.getActorOutput(32), mu_before, atol=1e-4)
    np.testing.assert_array_equal(dqn.getActorOutput(32), dqn.debug_read("actor_out"))
    np.testing.assert_allclose(dqn.getActorOutput(8, net=pkg.ACTOR_TARGET), mu_t[:8], atol=1e-4)
    with pytest.raises(pkg.DQNFatal):
        dqn.getActorOutput(33)
    dqn.close(); orc.close()


def test_sample_action_distribution(pkg, gpu):
    """SampleAction draws from discrete_distribution{max(0,dash+1), max(0,turn+1), 0, max(0,kick+1)}
    (src/dqn.cpp:181-186): chi-square against those probabilities; TACKLE never; arguments by offset."""
    dqn = pkg.DQN(59, minibatch=32, hidden=(64,), memory=100, seed=9)
    ao = np.array([0.5, -0.25, 0.9, -2.0, 11, 12, 13, 14, 15, 16], np.float32)    # kick logit -2 -> probability 0
    p = np.array([1.5, 0.75, 0.0, 0.0]); p /= p.sum()
    n = 20000
    acts = [dqn.SampleAction(ao) for _ in range(n)]
    cnt = np.bincount([a.action for a in acts], minlength=4)
    assert cnt[pkg.TACKLE] == 0 and cnt[pkg.KICK] == 0
    chi2 = sum((cnt[i] - n * p[i]) ** 2 / (n * p[i]) for i in (0, 1))
    assert chi2 < 10.83                                                  # 1 dof, p = 0.001
    for a in acts[:50]:
        assert (a.arg1, a.arg2) == ((11.0, 12.0) if a.action == pkg.DASH else (13.0, 0.0))
    dqn.close()


@pytest.mark.parametrize("entry", ["blocking", "async", "graph", "phased"])
def test_non_finite_target_is_reported_by_every_entry_point(pkg, gpu, entry):
    """A reward of +inf makes the TD target non-finite: the reference dies on CHECK(std::isfinite(target))
    (src/dqn.cpp:898).  The flag is raised on the device by the kernel that forms the target, so the
    asynchronous / captured / phased forms report it too, at the first read of the statistics."""
    B = 32
    dqn, orc, data, rng = make_pair(pkg, B=B, S=59, hidden=(64, 64), n_replay=256, use_graph=(entry == "graph"))
    s, a, r, mc, nx, term = [x.copy() for x in data]
    r[:] = np.inf
    dqn.ClearReplayMemory()
    dqn.add_transitions_arrays(s, a, r, mc, nx, term)
    idx = rng.integers(0, 256, size=B)
    if entry == "blocking":
        with pytest.raises(pkg.DQNFatal, match="Target not finite"):
            dqn.UpdateActorCritic(idx)
    else:
        if entry == "phased":
            for ph in (0, 1, 2):
                dqn.update_phase(ph, idx if ph == 0 else None)
        else:
            dqn.update_async(idx)
        with pytest.raises(pkg.DQNFatal, match="Target not finite"):
            dqn.read_stats()
    dqn.close(); orc.close()


def test_non_finite_gradient_norm_skips_the_step(pkg, gpu):
    """An overflowed backward (here provoked with NaN weights in the actor head) must not poison m, v, w
    and the targets through clip/inf = 0, 0 * inf = NaN: the optimiser pass is skipped and counted."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(64, 64), n_replay=256)
    w = dqn.get_params(0)
    w_bad = w.copy(); w_bad[-20] = np.nan                     # one actionpara_layer weight
    dqn.set_params(0, w_bad)
    m0, v0, wc0 = dqn.get_params(0, 1), dqn.get_params(0, 2), dqn.get_params(1)
    dqn.update_async(rng.integers(0, 256, size=32))
    with pytest.raises(pkg.DQNFatal, match="not finite"):
        dqn.read_stats()
    assert dqn.skipped_steps() >= 1
    np.testing.assert_array_equal(dqn.get_params(0, 1), m0)  # actor Adam history untouched
    np.testing.assert_array_equal(dqn.get_params(0, 2), v0)
    assert np.isfinite(dqn.get_params(1)).all()               # the critic step (finite gradients) did run
    assert not np.array_equal(dqn.get_params(1), wc0)
    dqn.close(); orc.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_update_returns_the_previous_updates_scalars(pkg, gpu, use_graph):
    """dqnhip_update_pipelined = dqnhip_update with the (loss, avg_q) read-back one update late: the same indices give
    the same weights bit for bit, call t returns what the blocking form returned at call t-1 ((0, 0) first), and
    dqnhip_read_stats drains the last one.  A non-finite target surfaces one call later."""
    B = 32
    kw = dict(B=B, S=59, hidden=(128, 64, 64, 64), n_replay=1024, use_graph=use_graph)
    d1, o1, data, rng = make_pair(pkg, **kw)
    d2, o2, _, _ = make_pair(pkg, **kw)
    idx = rng.integers(0, 1024, size=(6, B))
    blocking = [d1.UpdateActorCritic(i) for i in idx]
    piped = [d2.UpdateActorCriticPipelined(i) for i in idx]
    assert piped[0] == (0.0, 0.0)
    assert piped[1:] == blocking[:-1]
    assert d2.read_stats() == blocking[-1]
    for net in range(4):
        np.testing.assert_array_equal(d1.get_params(net), d2.get_params(net))
    assert d2.actor_iter() == 6
    # device-side sampling through the pipelined form, and the blocking benchmark of both forms
    d2.UpdateActorCriticPipelined(None)
    assert d2.BenchmarkBlocking(20, 2, seed=3, pipelined=False) > 0 and d2.BenchmarkBlocking(20, 2, seed=3, pipelined=True) > 0
    assert all(np.isfinite(d2.read_stats()))
    # CHECK(isfinite(target)) one call late
    s, a, r, mc, nx, term = data
    r = r.copy(); r[:] = np.inf
    d2.ClearReplayMemory(); d2.add_transitions_arrays(s, a, r, mc, nx, term)
    d2.UpdateActorCriticPipelined(idx[0])              # enqueued; reports the (finite) previous update
    with pytest.raises(pkg.DQNFatal, match="Target not finite"):
        d2.UpdateActorCriticPipelined(idx[1])
    for x in (d1, d2, o1, o2):
        x.close()


def test_pipelined_update_reports_a_skipped_step_exactly_once(pkg, gpu):
    """update t's read-back is enqueued before the call that reports update t-1's sticky flag clears it, so the
    next call used to fail a second time with 'update flags raised' although nothing new had happened."""
    B = 32
    dqn, orc, data, rng = make_pair(pkg, B=B, S=59, hidden=(64, 64), n_replay=256)
    idx = rng.integers(0, 256, size=(5, B))
    w = dqn.get_params(0)
    w_bad = w.copy(); w_bad[-20] = np.nan
    dqn.UpdateActorCriticPipelined(idx[0])                 # clean
    dqn.read_stats()
    dqn.set_params(0, w_bad)
    dqn.UpdateActorCriticPipelined(idx[1])                 # enqueues the update whose actor step is skipped
    dqn.set_params(0, w)                                   # (blocks: the weights are healthy again for what follows)
    with pytest.raises(pkg.DQNFatal, match="not finite"):
        dqn.UpdateActorCriticPipelined(idx[2])             # reports update 1's flag (update 2 itself is clean)
    assert dqn.skipped_steps() == 1
    dqn.UpdateActorCriticPipelined(idx[3])                 # must NOT report it again
    dqn.UpdateActorCriticPipelined(idx[4])
    assert all(np.isfinite(dqn.read_stats()))
    dqn.close(); orc.close()


def test_apply_update_reports_a_skipped_step(pkg, gpu):
    """dqnhip_read_stats reads the host-mapped words a full update's last block writes; an optimiser pass
    outside an update (dqnhip_apply_update) that skips on a non-finite norm must surface through them as well."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(64, 64), n_replay=256)
    dqn.UpdateActorCritic(rng.integers(0, 256, size=32))
    g = dqn.get_params(1, pkg.KIND_G)
    g[3] = np.inf
    dqn.set_params(1, g, pkg.KIND_G)
    w0 = dqn.get_params(1)
    dqn.apply_update(pkg.CRITIC)
    with pytest.raises(pkg.DQNFatal, match="Gradient norm not finite"):
        dqn.read_stats()
    assert dqn.skipped_steps() == 1
    np.testing.assert_array_equal(dqn.get_params(1), w0)
    assert all(np.isfinite(dqn.read_stats()))              # reported once, cleared
    dqn.close(); orc.close()
