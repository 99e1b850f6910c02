"""The optimiser pass in isolation: identical (w, g, m, v, w', iter) loaded into the HIP learner and the C oracle,
ONE Solver::ApplyUpdate each (dqnhip_apply_update / orc_apply_update: ClipGradients + Adam + Net::Update + soft
target update, src/dqn.cpp:904 tail, :964-970), results compared to a few ulp.  This replaces "within one Adam
step" of the whole-update tests as the pin for the optimiser: there the two sides already start from gradients
that differ by summation order; here the gradient is the same bits.

What may legitimately differ: the clip norm (oracle: one double sum per blob, added in float; HIP: fp32 partials
per block, tree-added) -> the clip scale agrees to ~1e-7 relative, so a clipped g, hence m, moves by <= 1-2 ulp
and v by <= 2-3 ulp; everything after that is the same sequence of fp32 operations (fma forms included)."""
import numpy as np
import pytest

from oracle import c_oracle, torch_ref

pytestmark = pytest.mark.gpu

ULP = np.finfo(np.float32).eps      # 2^-23


def _pair(pkg, S, hidden, precision="fp32", **kw):
    B = 128 if precision == "fp16" else 32
    d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=64, seed=1, precision=precision, **kw)
    okw = {}
    if "clip_grad" in kw: okw["clip"] = kw["clip_grad"]
    if "soft_update_freq" in kw: okw["soft_update_freq"] = kw["soft_update_freq"]
    if "tau" in kw: okw["tau"] = kw["tau"]
    o = c_oracle.Oracle(B=B, S=S, hidden=hidden, capacity=64, **okw)
    return d, o


def _load(pkg, d, o, rng, S, hidden, net, gnorm, it_a, it_c):
    actor = net == 0
    n = d.param_count(net)
    w = (torch_ref.init_params_np(rng, S, hidden, actor) * 3 + rng.normal(0, 1e-3, n)).astype(np.float32)
    wt = (w + rng.normal(0, 1e-2, n)).astype(np.float32)
    g = rng.normal(0, 1, n).astype(np.float32)
    g *= np.float32(gnorm / np.linalg.norm(g.astype(np.float64)))
    g[rng.integers(0, n, n // 50)] = 0.0                          # exact zeros (pad-like entries) stay exact
    m = rng.normal(0, 1e-3, n).astype(np.float32)
    v = (rng.uniform(0, 1, n) ** 4 * 1e-4).astype(np.float32)    # down to ~0: sqrt(v) + eps matters
    for x in (d, o):
        x.set_params(net, w); x.set_params(net + 2, wt)
        x.set_params(net, m, pkg.KIND_M); x.set_params(net, v, pkg.KIND_V)
        x.set_iters(it_a, it_c)
    d.set_params(net, g, pkg.KIND_G)
    o.grad_view(net)[:] = g
    return w, wt, g, m, v


def _check(pkg, d, o, net, before, exact_scale, t, clip=10.0):
    """exact_scale: the clip does not rescale the gradient (norm below the threshold, or clipping disabled) — the two
    sides then run the same fp32 operations on the same bits: m and v must be IDENTICAL, w within one rounding.
    Otherwise "a few ulp" is measured against the magnitude of the TERMS of each sum (m = (1-b1) g + b1 m0 can
    cancel, and an ulp of the result is then not the scale of the round-off)."""
    w0, wt0, g, m0, v0 = before
    b1, b2, eps = 0.95, 0.999, 1e-8
    lr = 1e-5 if net == 0 else 1e-3
    corr = np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    nrm = np.linalg.norm(g.astype(np.float64))
    gs = np.abs(g.astype(np.float64)) * (clip / nrm if (clip >= 0 and nrm > clip) else 1.0)
    m_terms = (1 - b1) * gs + b1 * np.abs(m0)
    ma, mb = d.get_params(net, pkg.KIND_M).astype(np.float64), o.get_params(net, pkg.KIND_M).astype(np.float64)
    va, vb = d.get_params(net, pkg.KIND_V).astype(np.float64), o.get_params(net, pkg.KIND_V).astype(np.float64)
    if exact_scale:
        np.testing.assert_array_equal(ma, mb); np.testing.assert_array_equal(va, vb)
    assert (np.abs(ma - mb) <= 3 * ULP * m_terms).all(), (np.abs(ma - mb) / np.maximum(m_terms, 1e-300)).max() / ULP
    assert (np.abs(va - vb) <= 6 * ULP * vb).all(), (np.abs(va - vb) / np.maximum(vb, 1e-300)).max() / ULP
    wa, wb = d.get_params(net), o.get_params(net)
    # the step itself (w0 - w1) against the scale of ITS terms, plus one rounding of w
    step_a, step_b = w0.astype(np.float64) - wa, w0.astype(np.float64) - wb
    step_scale = lr * corr * m_terms / (np.sqrt(vb) + eps)
    tol = (2 if exact_scale else 10) * ULP * step_scale + 1.0 * ULP * np.abs(wb)
    assert (np.abs(step_a - step_b) <= tol).all(), (np.abs(step_a - step_b) / np.maximum(tol, 1e-300)).max()
    assert np.abs(step_b).max() > 0
    ta, tb = d.get_params(net + 2), o.get_params(net + 2)
    assert (np.abs(ta.astype(np.float64) - tb) <= 1.0 * ULP * np.abs(tb) + np.abs(step_a - step_b)).all()
    return wa, wb


@pytest.mark.parametrize("net", [0, 1])
@pytest.mark.parametrize("t", [1, 2, 1000])
@pytest.mark.parametrize("gnorm,clip", [(3.0, 10.0), (250.0, 10.0), (250.0, -1.0)])     # clip inactive / active / disabled
def test_apply_update_matches_oracle_to_a_few_ulp(pkg, gpu, net, t, gnorm, clip):
    S, hidden = 59, (256, 128, 64, 64)
    d, o = _pair(pkg, S, hidden, clip_grad=clip)
    rng = np.random.default_rng(100 * net + t + int(gnorm))
    it = t - 1
    before = _load(pkg, d, o, rng, S, hidden, net, gnorm, it if net == 0 else it + 3, it if net == 1 else it + 5)
    d.apply_update(net); o.apply_update(net)
    _check(pkg, d, o, net, before, exact_scale=(clip < 0 or gnorm <= clip), t=t, clip=clip)
    assert (d.actor_iter(), d.critic_iter()) == tuple(o.get_iters())
    d.close(); o.close()


def test_apply_update_soft_update_schedule_and_zero_gradient(pkg, gpu):
    """soft_update_freq = 3: the target moves only when max_iter() (after both increments of a full update) is a
    multiple of 3 (src/dqn.cpp:967); a zero gradient with zero history leaves the weights bit-identical."""
    S, hidden = 59, (128, 64)
    d, o = _pair(pkg, S, hidden, soft_update_freq=3, tau=0.25)
    rng = np.random.default_rng(1)
    for it, moves in ((0, False), (1, False), (2, True), (5, True), (6, False)):
        before = _load(pkg, d, o, rng, S, hidden, 1, 5.0, it, it)
        d.apply_update(1); o.apply_update(1)
        _check(pkg, d, o, 1, before, exact_scale=True, t=it + 1)
        moved = not np.array_equal(d.get_params(3), before[1])
        assert moved == moves, (it, moved)
    n = d.param_count(0)
    w = torch_ref.init_params_np(rng, S, hidden, True)
    d.set_params(0, w); d.set_params(0, np.zeros(n, np.float32), pkg.KIND_G)
    d.set_params(0, np.zeros(n, np.float32), pkg.KIND_M); d.set_params(0, np.zeros(n, np.float32), pkg.KIND_V)
    d.apply_update(0)
    np.testing.assert_array_equal(d.get_params(0), w)
    assert d.skipped_steps() == 0
    d.close(); o.close()


def test_apply_update_is_the_pass_inside_an_update(pkg, gpu):
    """Phases 0 + 1 of an update leave the actor's gradient in its arena; finishing with phase 2 and finishing a twin
    learner with dqnhip_apply_update(ACTOR) give the same weights bit for bit (same kernel, norm taken by k_sumsq
    instead of the epilogue partials: the clip is inactive here, so the scale is exactly 1 either way)."""
    from synth import synth_replay
    S, hidden, B = 59, (128, 64, 64), 32
    rng = np.random.default_rng(2)
    w = [torch_ref.init_params_np(rng, S, hidden, a) for a in (True, False)]
    data = synth_replay(rng, 512, S, mean_len=10)
    idx = rng.integers(0, 512, B)
    ds = [pkg.DQN(S, minibatch=B, hidden=hidden, memory=1024, seed=1, clip_grad=1e9) for _ in range(2)]
    for d in ds:
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*data)
        d.update_phase(0, idx); d.update_phase(1)
    ds[0].update_phase(2)
    ds[1].update_abort()                       # the phased update is abandoned after phase 1 ...
    ds[1].apply_update(pkg.ACTOR)              # ... and the actor step applied on its own
    for net in (0, 2):
        np.testing.assert_array_equal(ds[0].get_params(net), ds[1].get_params(net))
    np.testing.assert_array_equal(ds[0].get_params(0, pkg.KIND_M), ds[1].get_params(0, pkg.KIND_M))
    np.testing.assert_array_equal(ds[0].get_params(0, pkg.KIND_V), ds[1].get_params(0, pkg.KIND_V))
    for d in ds:
        d.close()


def test_apply_update_fp16_learner_keeps_its_mirrors(pkg, gpu):
    """fp16 learner: the pass also writes the fp16 weight mirrors the GEMMs read — an update after an isolated
    ApplyUpdate must see the stepped weights (checked through the greedy action of the stepped actor)."""
    S, hidden = 58, (128, 128)
    d, o = _pair(pkg, S, hidden, precision="fp16")
    rng = np.random.default_rng(3)
    before = _load(pkg, d, o, rng, S, hidden, 0, 50.0, 0, 0)
    d.apply_update(0); o.apply_update(0)
    _check(pkg, d, o, 0, before, exact_scale=False, t=1)
    st = rng.uniform(-1, 1, (8, S)).astype(np.float32)
    a = d.SelectActionGreedily(st)             # fp32 acting path on the stepped master weights
    np.testing.assert_allclose(a, o.actor_forward(st), atol=1e-3, rtol=1e-3)
    d.close(); o.close()


@pytest.mark.parametrize("net", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("gnorm,clip", [(3.0, 10.0), (250.0, 10.0)])     # clip inactive / active
def test_sharded_optimiser_step_equals_the_replicated_one(pkg, gpu, net, world, gnorm, clip):
    """DQNHIP_DP_SHARD_OPT on ONE GPU: dqnhip_apply_update_sharded runs, for every rank of a `world`-rank group in turn, exactly
    what that rank runs between the reduce-scatter and the all-gather — the sum of squares of its slice of the (already
    reduced) gradient, the clip norm from the slice sums in rank order, clip + Adam + soft update on its slice — and must
    reproduce dqnhip_apply_update (the replicated form: every rank on the whole arena): bit for bit when the clip does not
    rescale (or world = 1: same summation tree), to a few ulp of the terms when it does (the norm is summed in another order)."""
    S, hidden = 59, (256, 128, 64, 64)
    a, oa = _pair(pkg, S, hidden, clip_grad=clip)
    b, ob = _pair(pkg, S, hidden, clip_grad=clip)
    t = 3
    before = _load(pkg, a, oa, np.random.default_rng(7 + net), S, hidden, net, gnorm, t - 1, t - 1)
    _load(pkg, b, ob, np.random.default_rng(7 + net), S, hidden, net, gnorm, t - 1, t - 1)
    a.apply_update(net); b.apply_update_sharded(net, world); ob.apply_update(net)
    exact = gnorm <= clip or world == 1
    if exact:
        for kind in (pkg.KIND_W, pkg.KIND_M, pkg.KIND_V):
            np.testing.assert_array_equal(a.get_params(net, kind), b.get_params(net, kind))
        np.testing.assert_array_equal(a.get_params(net + 2), b.get_params(net + 2))
    _check(pkg, b, ob, net, before, exact_scale=gnorm <= clip, t=t, clip=clip)        # and against the oracle, like the replicated form
    assert (a.actor_iter(), a.critic_iter()) == (b.actor_iter(), b.critic_iter())
    with pytest.raises(pkg.DQNFatal, match="divisible"):
        b.apply_update_sharded(net, 7)
    for x in (a, b, oa, ob):
        x.close()


@pytest.mark.parametrize("precision,half", [("fp32", False), ("fp16", False), ("fp16", True)])
@pytest.mark.parametrize("use_graph,clip", [(False, -1.0), (True, -1.0), (True, 10.0)])
def test_native_rccl_sharded_optimiser_one_rank(pkg, gpu, precision, half, use_graph, clip):
    """The sharded form through the real communicator with ONE rank (reduce-scatter / all-reduce / all-gather are identities):
    the slice is the whole arena and the clip norm is folded by the same tree, so the result is the REPLICATED data-parallel
    update's bit for bit (fp32, fp16, fp16 with the bf16 exchange), eager and as one captured hipGraph; the Adam history is complete
    after dqnhip_dp_gather_state."""
    from synth import synth_replay
    B, S, hid = 128, 59, (256, 128, 128)
    rng = np.random.default_rng(6)
    w = [torch_ref.init_params_np(rng, S, hid, act) * 5 for act in (True, False)]
    data = synth_replay(rng, 1024, S, mean_len=10)
    ds = [pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2, dp_world=1, dp_rank=0, precision=precision, use_graph=use_graph,
                  clip_grad=clip) for _ in range(2)]
    for d in ds:
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*data)
    ds[0].dp_init(pkg.DQN.dp_unique_id(), half_grads=half, shard_opt=True)
    ds[1].dp_init(pkg.DQN.dp_unique_id(), half_grads=half)            # the twin: the replicated form
    # clip disabled: the two forms run the same operations on the same bits.  clip = 10 (active at these weights): a one-rank
    # replicated group takes its norm from the GEMM epilogues' partial sums, the sharded one from k_sumsq over its slice —
    # two summation orders of the same number, so the clip scale agrees to ~1e-7 and everything after it to a few ulp
    exact = clip < 0
    for u in range(4):
        ds[0].dp_update(None); ds[1].dp_update(None)
        s0, s1 = ds[0].read_stats(), ds[1].read_stats()
        # (fp16 learner with the clip active: a weight that the ~1e-7 difference of the two clip scales moves across an fp16 rounding
        # boundary changes by 1e-3 of itself in the mirror the GEMMs read — measured 5e-5 on avg_q by the fourth update)
        assert s0 == s1 if exact else np.allclose(s0, s1, rtol=1e-5 if precision == "fp32" else 3e-4, atol=1e-7), (u, s0, s1)
    assert ds[0].dp_graph_active() == use_graph
    ds[0].dp_gather_state()
    # (fp16 with the clip active, see above: a few weights sit on the other side of an fp16 rounding boundary in the mirrors -> 1e-4-level
    # relative differences in what four updates made of them)
    rt, at = (2e-5, 1e-7) if precision == "fp32" else (5e-4, 2e-6)
    same = np.testing.assert_array_equal if exact else (lambda a, b: np.testing.assert_allclose(a, b, rtol=rt, atol=at))
    for net in range(4):
        same(ds[0].get_params(net), ds[1].get_params(net))
    for kind in (pkg.KIND_M, pkg.KIND_V):
        for net in (0, 1):
            same(ds[0].get_params(net, kind), ds[1].get_params(net, kind))
    assert ds[0].skipped_steps() == 0 and ds[0].actor_iter() == 4
    for call in (lambda: ds[0].update_async(None), lambda: ds[0].update_phase(0, None)):
        with pytest.raises(pkg.DQNFatal, match="dqnhip_dp_update"):
            call()
    ds[0].dp_update(None)
    with pytest.raises(pkg.DQNFatal, match="dqnhip_dp_gather_state"):
        ds[0].dp_destroy()                                       # never an implicit collective at teardown: it is asked for
    ds[0].dp_gather_state()
    ds[0].dp_destroy()
    ds[0].UpdateActorCritic(rng.integers(0, 1024, B))           # a plain learner again
    for d in ds:
        d.close()
