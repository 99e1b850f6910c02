"""dqnhip_update_async_n: n updates with on-device sampling in one call — the reference's inner loops
`for (i < n_updates) dqn->Update()` (src/dqn_main.cpp:359-361) and DQN::Benchmark (src/dqn.cpp:487-498).  With use_graph
the updates are replayed sixteen to a hipGraph launch (the gather and the first layers of update u + 1 riding in update u's two optimiser launches); whatever the grouping, the state must be exactly what n single calls
leave."""
import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu


def _state(d, pkg):
    out = [d.get_params(n) for n in range(4)]
    out += [d.get_params(n, k) for n in (0, 1) for k in (pkg.KIND_M, pkg.KIND_V)]
    return out, d.read_stats(), (d.actor_iter(), d.critic_iter())


def _mk(pkg, use_graph, precision="fp32", B=64, hidden=(256, 128, 64, 64)):
    d = pkg.DQN(59, minibatch=B, hidden=hidden, memory=4096, seed=11, use_graph=use_graph, precision=precision)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 3000, 59))
    return d


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("precision,B,hidden", [("fp32", 64, (256, 128, 64, 64)), ("fp32", 256, (1024, 1024, 1024, 1024)),
                                                ("fp16", 128, (256, 128, 128))])
@pytest.mark.parametrize("n", [0, 1, 15, 16, 35, 48])
def test_n_updates_equal_n_single_calls(pkg, gpu, use_graph, precision, B, hidden, n):
    if B == 256 and n not in (16, 35):
        pytest.skip("the BASELINE tower only for the grouped cases")
    a = _mk(pkg, use_graph, precision, B, hidden)
    for _ in range(n):
        a.update_async(None)
    sa = _state(a, pkg); a.close()
    b = _mk(pkg, use_graph, precision, B, hidden)
    b.update_async_n(n)
    sb = _state(b, pkg); b.close()
    assert sa[2] == sb[2] == (n, n)
    assert sa[1] == sb[1]
    for x, y in zip(sa[0], sb[0]):
        np.testing.assert_array_equal(x, y)


def test_n_updates_interleave_with_other_entry_points(pkg, gpu):
    """a grouped replay between single updates, explicit-index updates, new transitions and a parameter write: the device-side
    sampling counters and the ring state carry over"""
    rng = np.random.default_rng(4)
    extra = synth_replay(np.random.default_rng(9), 500, 59)
    res = []
    for grouped in (False, True):
        d = _mk(pkg, True)
        d.update_async(None)
        d.UpdateActorCritic(np.arange(64, dtype=np.int32))
        (d.update_async_n(37) if grouped else [d.update_async(None) for _ in range(37)])
        d.add_transitions_arrays(*extra)
        w = d.get_params(0); d.set_params(0, w * 1.01)
        (d.update_async_n(19) if grouped else [d.update_async(None) for _ in range(19)])
        res.append(_state(d, pkg)); d.close()
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] == (58, 58)
    for x, y in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(x, y)


def test_n_updates_refusals(pkg, gpu):
    d = pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=256, seed=1)
    with pytest.raises(RuntimeError, match="replay memory is empty"):
        d.update_async_n(3)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 100, 59))
    with pytest.raises(RuntimeError, match="n must be >= 0"):
        d.update_async_n(-1)
    d.update_phase(0, None)
    with pytest.raises(RuntimeError, match="phased update is in progress"):
        d.update_async_n(2)
    d.update_abort()
    d.update_async_n(2)
    assert d.actor_iter() == 2
    d.close()
    g = pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=256, seed=1, dp_world=1, dp_rank=0)
    g.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 100, 59))
    g.dp_init(pkg.DQN.dp_unique_id(), half_grads=True)
    with pytest.raises(RuntimeError, match="dqnhip_dp_update"):
        g.update_async_n(2)
    g.close()


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("precision,flags", [("fp32", {}), ("fp32", {"per_layer": True}), ("fp32", {"shard_opt": True}),
                                             ("fp16", {"half_grads": True}), ("fp16", {"half_grads": True, "shard_opt": True})])
def test_dp_n_updates_equal_n_single_calls(pkg, gpu, use_graph, precision, flags):
    """dqnhip_dp_update_n on a one-rank RCCL group (every exchange form): sixteen updates — collectives included — per
    hipGraph launch, each gather riding in the previous update's last launch, against single dqnhip_dp_update calls."""
    B, hidden = (128, (256, 128, 128)) if precision == "fp16" else (64, (256, 128, 64, 64))
    st = []
    for grouped in (False, True):
        d = pkg.DQN(59, minibatch=B, hidden=hidden, memory=4096, seed=11, use_graph=use_graph, precision=precision, dp_world=1, dp_rank=0)
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 3000, 59))
        d.dp_init(pkg.DQN.dp_unique_id(), **flags)
        d.dp_update(None)
        (d.dp_update_n(37) if grouped else [d.dp_update(None) for _ in range(37)])
        if use_graph:
            assert d.dp_graph_active()
        if flags.get("shard_opt"):
            d.dp_gather_state()
        st.append(_state(d, pkg)); d.close()
    assert st[0][1] == st[1][1] and st[0][2] == st[1][2] == (38, 38)
    for x, y in zip(st[0][0], st[1][0]):
        np.testing.assert_array_equal(x, y)
