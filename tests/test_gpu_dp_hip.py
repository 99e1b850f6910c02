"""The HIP learner's dp_world > 1 arithmetic, checked numerically on ONE GPU.

Two learners (dp_rank 0 / 1 of a dp_world = 2 group) live on device 0, each with its own replay
shard and half of the minibatch; dqnhip_reduce_gradients_local stands in for the all-reduce.  What
runs is exactly what a 2-GPU job runs between the collectives: the global-B EuclideanLoss
normaliser, the un-normalised actor gradient sum (src/dqn.cpp:918-921), k_sumsq + clip on the
REDUCED gradient, the [loss_sum, q_sum] tails.  Compared with (a) ONE HIP learner fed the
concatenated minibatch, (b) the C oracle at the global batch; the two ranks must stay bit-identical.
Also: the native RCCL path (dqnhip_dp_*) with a 1-rank communicator.
"""
import numpy as np
import pytest

from oracle import c_oracle, torch_ref
from synth import synth_replay

pytestmark = pytest.mark.gpu

N_SHARD = 1024


def _fro(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _group(pkg, Bl, S, hid, wscale, precision, seed=3, world=2):
    rng = np.random.default_rng(seed)
    w = [torch_ref.init_params_np(rng, S, hid, a) * wscale for a in (True, False)]
    shards = [synth_replay(np.random.default_rng(10 + r), N_SHARD, S, mean_len=10) for r in range(world)]
    ranks = [pkg.DQN(S, minibatch=Bl, hidden=hid, memory=4096, seed=7, dp_world=world, dp_rank=r, precision=precision)
             for r in range(world)]
    one = pkg.DQN(S, minibatch=Bl * world, hidden=hid, memory=4 * N_SHARD + 1, seed=7, precision=precision)
    orc = c_oracle.Oracle(B=Bl * world, S=S, hidden=hid, capacity=4 * N_SHARD + 1)
    for d in ranks + [one]:
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
    for net in (0, 1):
        orc.set_params(net, w[net]); orc.clone_to_target(net)
    for r, d in enumerate(ranks):
        d.add_transitions_arrays(*shards[r])
    for sh in shards:                      # the single learner / oracle hold the concatenation, shard after shard
        one.add_transitions_arrays(*sh); orc.add_transitions(*sh)
    return ranks, one, orc, rng


def _dp_step(pkg, ranks, idx_local):
    for d, i in zip(ranks, idx_local):
        d.update_phase(0, i)
    pkg.reduce_gradients_local(ranks, pkg.CRITIC)
    g_c = [d.get_params(1, pkg.KIND_G) for d in ranks]
    for d in ranks:
        d.update_phase(1)
    pkg.reduce_gradients_local(ranks, pkg.ACTOR)
    g_a = [d.get_params(0, pkg.KIND_G) for d in ranks]
    for d in ranks:
        d.update_phase(2)
    return g_c, g_a


@pytest.mark.parametrize("shape", [
    dict(Bl=64, S=59, hid=(256, 128, 64, 64), wscale=5.0, tol_orc=1e-5),
    dict(Bl=32, S=77, hid=(128, 64), wscale=5.0, tol_orc=1e-5),                       # 2v1 state size (BASELINE config #4)
    dict(Bl=128, S=58, hid=(1024, 1024, 1024, 1024), wscale=2.0, tol_orc=1e-5),      # BASELINE shape split over two ranks
])
def test_hip_dp2_matches_single_learner_and_oracle(pkg, gpu, shape):
    Bl, S, hid = shape["Bl"], shape["S"], shape["hid"]
    ranks, one, orc, rng = _group(pkg, Bl, S, hid, shape["wscale"], "fp32")
    n_it = 2 if len(hid) == 4 and hid[0] == 1024 else 3
    for it in range(n_it):
        idx_local = [rng.integers(0, N_SHARD, size=Bl) for _ in ranks]
        idx_global = np.concatenate([i + r * N_SHARD for r, i in enumerate(idx_local)])
        # reference runs, phase by phase (gradients are complete at the phase boundaries)
        one.update_phase(0, idx_global); orc.update_phase(0, idx_global)
        gc_one, gc_orc = one.get_params(1, pkg.KIND_G), orc.grad_view(1).copy()
        one.update_phase(1); orc.update_phase(1, idx_global)
        ga_one, ga_orc = one.get_params(0, pkg.KIND_G), orc.grad_view(0).copy()
        one.update_phase(2); orc.update_phase(2, idx_global)
        g_c, g_a = _dp_step(pkg, ranks, idx_local)
        # the reduced gradients are the same bits on both ranks
        np.testing.assert_array_equal(g_c[0], g_c[1]); np.testing.assert_array_equal(g_a[0], g_a[1])
        # = the single learner's gradient on the concatenated minibatch (row sums in a different order)
        assert _fro(g_c[0], gc_one) <= 1e-5, _fro(g_c[0], gc_one)
        assert _fro(g_a[0], ga_one) <= 1e-5, _fro(g_a[0], ga_one)
        assert _fro(g_c[0], gc_orc) <= shape["tol_orc"], _fro(g_c[0], gc_orc)
        assert _fro(g_a[0], ga_orc) <= shape["tol_orc"], _fro(g_a[0], ga_orc)
        s0, s1, s_one, s_orc = ranks[0].read_stats(), ranks[1].read_stats(), one.read_stats(), orc.last_stats()
        assert s0 == s1                                              # (critic_loss, avg_q): all-reduced tails
        assert abs(s0[0] - s_one[0]) <= 1e-5 * max(1.0, abs(s_one[0])) and abs(s0[1] - s_one[1]) <= 1e-5 * max(1.0, abs(s_one[1]))
        assert abs(s0[0] - s_orc[0]) <= 1e-4 * max(1.0, abs(s_orc[0])) and abs(s0[1] - s_orc[1]) <= 1e-4 * max(1.0, abs(s_orc[1]))
        # each rank's rows are its half of the global minibatch
        q_one = one.debug_read("q_train")
        for r, d in enumerate(ranks):
            np.testing.assert_allclose(d.debug_read("q_train"), q_one[r * Bl:(r + 1) * Bl], rtol=1e-5, atol=1e-5)
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net in range(4):
        a, b = ranks[0].get_params(net), ranks[1].get_params(net)
        np.testing.assert_array_equal(a, b)                          # replicas never diverge
        d = np.abs(a - one.get_params(net))
        assert d.max() <= n_it * lr[net] + 1e-6 and d.mean() <= 0.01 * lr[net] + 1e-8, (net, d.max(), d.mean())
    for kind in (pkg.KIND_M, pkg.KIND_V):
        for net in (0, 1):
            np.testing.assert_array_equal(ranks[0].get_params(net, kind), ranks[1].get_params(net, kind))
            b = one.get_params(net, kind)
            np.testing.assert_allclose(ranks[0].get_params(net, kind), b, rtol=1e-3, atol=1e-5 * np.abs(b).max())
    assert ranks[0].actor_iter() == n_it and ranks[1].critic_iter() == n_it
    for d in ranks + [one]:
        d.close()
    orc.close()


def test_two_agents_with_a_two_rank_group_each(pkg, gpu):
    """BASELINE.json configs[3]'s layout (2v1: S = 77; two agents, each a 2-rank data-parallel group with its own
    replay shards) with all four learners on one GPU: the groups exchange nothing with each other (the reference's
    agents are independent DQNs, src/dqn_main.cpp:264), each group's ranks stay bit-identical and each group
    reproduces ITS single learner on the concatenated minibatch."""
    Bl, S, hid = 32, 77, (128, 64, 64)
    groups = [_group(pkg, Bl, S, hid, 5.0, "fp32", seed=3 + 10 * a) for a in range(2)]
    for it in range(3):
        for a, (ranks, one, orc, rng) in enumerate(groups):
            idx_local = [rng.integers(0, N_SHARD, size=Bl) for _ in ranks]
            idx_global = np.concatenate([i + r * N_SHARD for r, i in enumerate(idx_local)])
            one.update_phase(0, idx_global); gc_one = one.get_params(1, pkg.KIND_G)
            one.update_phase(1); ga_one = one.get_params(0, pkg.KIND_G)
            one.update_phase(2)
            # interleave the two agents' phases on the device: agent 0 phase p, agent 1 phase p, ... is what two
            # agent threads produce; here agent a runs to completion per iteration, the other agent's learners
            # sit between its collectives with their own gradient arenas
            g_c, g_a = _dp_step(pkg, ranks, idx_local)
            np.testing.assert_array_equal(g_c[0], g_c[1]); np.testing.assert_array_equal(g_a[0], g_a[1])
            assert _fro(g_c[0], gc_one) <= 1e-5 and _fro(g_a[0], ga_one) <= 1e-5
            assert ranks[0].read_stats() == ranks[1].read_stats()
    # different agents, different weights (different seeds): nothing leaked across the groups
    assert not np.array_equal(groups[0][0][0].get_params(0), groups[1][0][0].get_params(0))
    for ranks, one, orc, rng in groups:
        for net in range(4):
            np.testing.assert_array_equal(ranks[0].get_params(net), ranks[1].get_params(net))
            d = np.abs(ranks[0].get_params(net) - one.get_params(net))
            assert d.mean() <= 1e-7, (net, d.mean())
        for d in ranks + [one]:
            d.close()
        orc.close()


def test_hip_dp2_fp16_ranks_bit_identical(pkg, gpu):
    """Mixed-precision learner under data parallelism: the loss scale uses the GLOBAL batch; both
    ranks hold the same bits after every update and track one fp16 learner on the concatenation."""
    Bl, S, hid = 128, 59, (256, 128)
    ranks, one, orc, rng = _group(pkg, Bl, S, hid, 5.0, "fp16")
    for it in range(3):
        idx_local = [rng.integers(0, N_SHARD, size=Bl) for _ in ranks]
        idx_global = np.concatenate([i + r * N_SHARD for r, i in enumerate(idx_local)])
        one.update_phase(0, idx_global)
        gc_one = one.get_params(1, pkg.KIND_G)
        one.update_phase(1)
        ga_one = one.get_params(0, pkg.KIND_G)
        one.update_phase(2)
        g_c, g_a = _dp_step(pkg, ranks, idx_local)
        np.testing.assert_array_equal(g_c[0], g_c[1]); np.testing.assert_array_equal(g_a[0], g_a[1])
        # same fp16 rounding points; the split changes which rows share a 128-row tile, not the values
        # (first update: identical weights.  Later ones start from weights that differ by what the fp16 roundings of the first made of
        # the two summation orders; measured 2.02e-3 in update 2 / 3 since the head's dW is a column-sum block of the last backward launch)
        tol = 2e-3 if it == 0 else 3e-3
        print("fp16 dp2 update", it, _fro(g_c[0], gc_one), _fro(g_a[0], ga_one))
        assert _fro(g_c[0], gc_one) <= tol, (it, _fro(g_c[0], gc_one))
        assert _fro(g_a[0], ga_one) <= tol, (it, _fro(g_a[0], ga_one))
        assert ranks[0].read_stats() == ranks[1].read_stats()
    for net in range(4):
        np.testing.assert_array_equal(ranks[0].get_params(net), ranks[1].get_params(net))
    for d in ranks + [one]:
        d.close()
    orc.close()


def test_phase_order_and_index_count_are_checked(pkg, gpu):
    d = pkg.DQN(59, minibatch=32, hidden=(64,), memory=2048, dp_world=2, dp_rank=1)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(1), 512, 59, mean_len=10))
    with pytest.raises(pkg.DQNFatal, match="out of order"):
        d.update_phase(1)                       # phase 1 without phase 0: stale gradients
    with pytest.raises(pkg.DQNFatal, match="indices"):
        d.update_phase(0, np.arange(16))        # short index array: would be an out-of-bounds host read
    with pytest.raises(pkg.DQNFatal, match="requires dqnhip_update_phase"):
        d.update_async()
    d.update_phase(0, np.arange(32))
    with pytest.raises(pkg.DQNFatal, match="out of order"):
        d.update_phase(2)
    with pytest.raises(pkg.DQNFatal, match="out of order"):
        d.update_phase(0, np.arange(32))
    d.update_phase(1); d.update_phase(2)
    d.update_phase(10, np.arange(32))
    with pytest.raises(pkg.DQNFatal, match="out of order"):
        d.update_phase(1)
    d.update_phase(11); d.update_phase(1); d.update_phase(2)
    assert d.actor_iter() == 2
    d.close()


def test_dp_rank_decorrelates_sampling_not_init(pkg, gpu):
    """Same cfg.seed on every rank: identical gaussian initialisation, different sample streams."""
    ds = [pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=4096, seed=11, dp_world=2, dp_rank=r) for r in (0, 1)]
    for net in range(4):
        np.testing.assert_array_equal(ds[0].get_params(net), ds[1].get_params(net))
    data = synth_replay(np.random.default_rng(1), 2000, 59, mean_len=10)
    for d in ds:
        d.add_transitions_arrays(*data)
        d.update_phase(0)
    i0, i1 = ds[0].debug_read("idx").astype(np.int64), ds[1].debug_read("idx").astype(np.int64)
    np.testing.assert_array_equal(i0, c_oracle.philox_indices(11, 0, 32, 2000))      # rank 0 = the documented stream
    assert (i0 != i1).mean() > 0.9
    for d in ds:
        d.close()


@pytest.mark.parametrize("per_layer", [False, True])
def test_native_rccl_one_rank(pkg, gpu, per_layer):
    """dqnhip_dp_init / dqnhip_dp_update with a real RCCL communicator of one rank inside the library
    (no torch, no Python between the phases): a 1-rank sum all-reduce is the identity, so the result
    is bit-identical to the plain update; per-layer bucketing on the communication stream changes
    nothing either.  The init broadcast leaves rank 0's state in place."""
    B, S, hid = 64, 59, (256, 128, 64, 64)
    rng = np.random.default_rng(5)
    w = [torch_ref.init_params_np(rng, S, hid, a) * 5 for a in (True, False)]
    data = synth_replay(rng, 1024, S, mean_len=10)
    idx = rng.integers(0, 1024, size=(4, B))
    dp = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2, dp_world=1, dp_rank=0)
    ref = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2)
    for d in (dp, ref):
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*data)
    with pytest.raises(pkg.DQNFatal, match="no communicator"):
        dp.dp_update(idx[0])
    dp.dp_init(pkg.DQN.dp_unique_id(), per_layer=per_layer)
    with pytest.raises(pkg.DQNFatal, match="already has a communicator"):
        dp.dp_init(pkg.DQN.dp_unique_id())
    for net in range(4):
        np.testing.assert_array_equal(dp.get_params(net), ref.get_params(net))     # broadcast from rank 0 = itself
    for u in range(4):
        dp.dp_update(idx[u]); s_dp = dp.read_stats()
        s_ref = ref.UpdateActorCritic(idx[u])
        assert s_dp == s_ref
    for net in range(4):
        np.testing.assert_array_equal(dp.get_params(net), ref.get_params(net))
    dp.dp_update(None)                                   # on-device sampling through the native path
    assert all(np.isfinite(dp.read_stats()))
    dp.dp_broadcast_params(0)
    dp.close(); ref.close()


def test_native_rccl_file_rendezvous(pkg, gpu, tmp_path):
    """dqnhip_dp_init_file: a stale file of an earlier job is no obstacle (rank 0 clears it and waiters would only
    accept a file that carries their own nonce), the files are gone once the group is up, and the same path serves
    the next group."""
    path = tmp_path / "rccl_id"
    path.write_bytes(b"\x07" * 128)                       # left over from a crashed job
    (tmp_path / "rccl_id.req1").write_bytes(b"\x01" * 8)
    for _ in range(2):
        d = pkg.DQN(59, minibatch=32, hidden=(64,), memory=2048, dp_world=1, dp_rank=0)
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(1), 512, 59, mean_len=10))
        d.dp_init_file(str(path), timeout_s=10)
        assert not path.exists()
        d.dp_update(np.arange(32))
        assert all(np.isfinite(d.read_stats()))
        d.close()


def test_abandoned_phased_update_does_not_wedge_the_learner(pkg, gpu):
    """A caller whose exchange step failed abandons the update with dqnhip_update_abort; a phase that fails abandons
    it by itself.  Either way the next update starts normally (next_phase used to stay stuck)."""
    d = pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=2048, dp_world=2, dp_rank=0)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(1), 512, 59, mean_len=10))
    d.update_phase(0, np.arange(32)); d.update_phase(1)
    with pytest.raises(pkg.DQNFatal, match="out of order"):
        d.update_phase(0, np.arange(32))
    d.update_abort()
    d.update_phase(0, np.arange(32)); d.update_phase(1); d.update_phase(2)
    with pytest.raises(pkg.DQNFatal, match="out of range"):
        d.update_phase(0, np.arange(32) + 5000)          # fails inside phase 0
    d.update_phase(0, np.arange(32)); d.update_phase(1); d.update_phase(2)
    assert d.actor_iter() == 2 and all(np.isfinite(d.read_stats()))
    d.close()


def test_non_finite_target_on_one_rank_is_reported_by_all(pkg, gpu):
    """The TD-target flag is raised from a rank's OWN replay shard; it rides in the all-reduced critic tail so that
    every rank reports 'Target not finite!' for the same update (otherwise one rank stops and the others walk
    into the next collective)."""
    ranks = [pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=2048, seed=4, dp_world=2, dp_rank=r) for r in (0, 1)]
    for r, d in enumerate(ranks):
        s, a, rew, mc, nx, term = synth_replay(np.random.default_rng(1 + r), 512, 59, mean_len=10)
        if r == 1:
            rew[:] = np.inf
        d.add_transitions_arrays(s, a, rew, mc, nx, term)
    _dp_step(pkg, ranks, [np.arange(32), np.arange(32)])
    for d in ranks:
        with pytest.raises(pkg.DQNFatal, match="Target not finite"):
            d.read_stats()
    for d in ranks:
        d.close()


def _bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("precision,B,hid", [("fp32", 64, (256, 128, 64, 64)), ("fp16", 128, (256, 128))])
def test_native_rccl_half_grads_one_rank(pkg, gpu, precision, B, hid):
    """DQNHIP_DP_HALF_GRADS: the gradient arena crosses the (here: one-rank) all-reduce as bf16 and is widened again
    by the clip-norm pass; the [loss, q, flag] tails travel as fp32.  After the first update the critic's arena holds
    exactly bf16(plain gradient) (round-to-nearest-even), the reported scalars are those of the plain update, and
    three updates stay within the 8-bit gradient precision of the plain learner."""
    S = 59
    rng = np.random.default_rng(5)
    w = [torch_ref.init_params_np(rng, S, hid, a) * 5 for a in (True, False)]
    data = synth_replay(rng, 1024, S, mean_len=10)
    idx = rng.integers(0, 1024, size=(3, B))
    dp = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2, dp_world=1, dp_rank=0, precision=precision)
    ref = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2, precision=precision)
    for d in (dp, ref):
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*data)
    dp.dp_init(pkg.DQN.dp_unique_id(), half_grads=True)
    ref.update_phase(0, idx[0]); g_c = ref.get_params(1, pkg.KIND_G)
    ref.update_phase(1); g_a = ref.get_params(0, pkg.KIND_G); ref.update_phase(2)
    dp.dp_update(idx[0])
    np.testing.assert_array_equal(dp.get_params(1, pkg.KIND_G), _bf16(g_c))
    ga_dp = dp.get_params(0, pkg.KIND_G)
    np.testing.assert_array_equal(ga_dp, _bf16(ga_dp))                       # bf16 values ...
    assert _fro(ga_dp, g_a) <= 2e-2                                          # ... of (nearly) the plain actor gradient
    s_dp, s_ref = dp.read_stats(), ref.read_stats()
    assert abs(s_dp[0] - s_ref[0]) <= 1e-6 * max(1, abs(s_ref[0]))           # loss: same forward, fp32 tail
    assert abs(s_dp[1] - s_ref[1]) <= 2e-3 * max(1, abs(s_ref[1]))           # avg Q: after the (bf16-gradient) critic step
    for u in (1, 2):
        dp.dp_update(idx[u]); ref.UpdateActorCritic(idx[u])
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net in range(4):
        dd = np.abs(dp.get_params(net) - ref.get_params(net))
        assert dd.max() <= 3 * lr[net] + 1e-6 and dd.mean() <= 0.05 * lr[net] + 1e-8, (net, dd.max(), dd.mean())
    assert dp.skipped_steps() == 0
    # the bf16 exchange lives inside dqnhip_dp_update: the other update entry points refuse instead of reading a stale image
    for call in (lambda: dp.update_async(idx[0]), lambda: dp.update_phase(0, idx[0]), lambda: dp.UpdateActorCriticPipelined(idx[0])):
        with pytest.raises(pkg.DQNFatal, match="dqnhip_dp_update"):
            call()
    dp.dp_destroy()                                          # without the communicator it is a plain learner again
    dp.UpdateActorCritic(idx[0])
    dp.close(); ref.close()


@pytest.mark.parametrize("precision,per_layer,half", [("fp32", False, False), ("fp32", True, False), ("fp16", False, True)])
def test_native_rccl_update_replays_as_one_graph(pkg, gpu, precision, per_layer, half):
    """cfg.use_graph under native data parallelism: phase 0 / all-reduce / phase 1 / all-reduce / phase 2 — RCCL's
    kernels and the communication-stream fork/join included — are captured once and replayed.  A graph-replaying
    group member and an eager one hold the same bits after every update."""
    B, S, hid = 128, 59, (256, 128, 128)
    rng = np.random.default_rng(6)
    w = [torch_ref.init_params_np(rng, S, hid, a) * 5 for a in (True, False)]
    data = synth_replay(rng, 1024, S, mean_len=10)
    ds = [pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2, dp_world=1, dp_rank=0, precision=precision, use_graph=g)
          for g in (True, False)]
    for d in ds:
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*data)
        d.dp_init(pkg.DQN.dp_unique_id(), per_layer=per_layer, half_grads=half)
    for u in range(5):
        for d in ds:
            d.dp_update(None)                           # on-device sampling: same seed, same counter
        assert ds[0].read_stats() == ds[1].read_stats()
    assert ds[0].dp_graph_active(), "RCCL refused the capture: the data-parallel update runs eagerly"
    assert not ds[1].dp_graph_active()
    for net in range(4):
        np.testing.assert_array_equal(ds[0].get_params(net), ds[1].get_params(net))
    assert ds[0].actor_iter() == ds[1].actor_iter() == 5
    ds[0].dp_update(rng.integers(0, 1024, B))           # explicit indices: eager path beside the captured graph
    assert all(np.isfinite(ds[0].read_stats()))
    for d in ds:
        d.close()
