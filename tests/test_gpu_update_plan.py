"""dqnhip_get_update_plan: the merged forms a learner's update takes and the kernels it launches, counted from a capture of the
very sequence dqnhip_update* enqueues.  The merged launches are gated by shape predicates (learner.hip, plan_of); one that
silently stops matching at a BASELINE shape would pass every parity test and only show up as a slower bench — here it is a
failed assertion.  Reference work covered by these launches: DQN::UpdateActorCritic, src/dqn.cpp:828-972."""
import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu

TOWER = (1024, 1024, 1024, 1024)
REF_TOWER = (1024, 512, 256, 128)
FP32_ALL = {"bwd_shifted_critic", "bwd_shifted_actor", "head_wgrad_rides_critic", "head_wgrad_rides_actor", "q_train_in_dgrad",
            "head_seed_fused", "dqda_head_bwd", "critic_l0_rides", "first_layers_merged", "early_gather_l0"}

# (name, constructor arguments, data-parallel flags or None, forms, (single, graph_first, in_graph) launches)
CASES = [
    # BASELINE configs[1]: every merged form, 26 launches stand-alone, 24 inside a sixteen-update graph (gather and first layers ride)
    ("configs1_b256_4x1024", dict(state_size=58, minibatch=256, hidden=TOWER), None, FP32_ALL, (26, 26, 24)),
    # the reference's compile-time defaults (configs[0] on the GPU): layers narrower than 512 take the pair launches, not the shifted
    # schedule -> no k_dgrad_qtrain (it is the shifted schedule's first launch); k_dqda_head_bwd takes any tower-top width since round 6
    ("configs0_b32_ref_tower", dict(state_size=59, minibatch=32, hidden=REF_TOWER), None,
     {"head_wgrad_rides_critic", "head_wgrad_rides_actor", "head_seed_fused", "dqda_head_bwd", "critic_l0_rides", "first_layers_merged", "early_gather_l0"}, (27, 27, 25)),
    # configs[2]'s learner (S = 68: both first panels are 128 wide — the actor's first-layer riders of k_adam_soft_l0 are built for 64)
    ("configs2_b256_s68", dict(state_size=68, minibatch=256, hidden=TOWER), None, FP32_ALL - {"early_gather_l0"}, (26, 26, 25)),
    # configs[3] / weak scaling: a rank of a replicated data-parallel group at 256 rows runs the same merged forms (round 6)
    ("dp_rank_b256_half_grads", dict(state_size=58, minibatch=256, hidden=TOWER), dict(half_grads=True), FP32_ALL | {"data_parallel", "dp_tails_ride"}, (28, 28, 26)),
    # ... the per-rank shape of a 4096-row minibatch on 8 GPUs, fp32
    ("dp_rank_b512_half_grads", dict(state_size=58, minibatch=512, hidden=TOWER), dict(half_grads=True), FP32_ALL | {"data_parallel", "dp_tails_ride"}, (28, 28, 26)),
    # configs[4] on one GPU, fp32: 4096 rows take the big head kernels; no rider fits
    ("configs4_b4096_fp32", dict(state_size=58, minibatch=4096, hidden=TOWER), None,
     {"bwd_shifted_critic", "bwd_shifted_actor", "head_seed_fused"}, (33, 33, 32)),
]


def _make(pkg, kw, dp):
    d = pkg.DQN(memory=8192, seed=3, use_graph=True, **kw)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 4096, kw["state_size"]))
    if dp is not None:
        d.dp_init(pkg.DQN.dp_unique_id(), **dp)
    return d


@pytest.mark.parametrize("name,kw,dp,forms,launches", CASES, ids=[c[0] for c in CASES])
def test_plan_at_named_shapes(pkg, gpu, name, kw, dp, forms, launches):
    d = _make(pkg, kw, dp)
    p = d.update_plan()
    print(name, p)
    assert set(p["forms"]) == forms, (name, sorted(set(p["forms"]) ^ forms))
    assert p["updates_per_graph"] == 16
    assert p["collectives"] == (0 if dp is None else 3 if dp.get("half_grads") else 2)
    assert 0 < p["launches_in_graph"] <= p["launches_graph_first"] <= p["launches_single"] + 1
    if launches is not None:
        assert (p["launches_single"], p["launches_graph_first"], p["launches_in_graph"]) == launches, (name, p)
    # counting is a capture that is thrown away: the learner then updates as if nothing had happened, eager / one graph / sixteen
    if dp is None:
        d.update_async_n(17)
    else:
        d.dp_update_n(17)
    loss, q = d.read_stats()
    assert np.isfinite(loss) and np.isfinite(q)
    assert d.update_plan() == p
    d.close()


@pytest.mark.parametrize("B", [512, 4096])
def test_plan_fp16(pkg, gpu, B):
    """configs[4]: the fp16 learner at its one-GPU minibatch and at its per-rank shape on 8 GPUs."""
    d = pkg.DQN(58, minibatch=B, hidden=TOWER, memory=8192, seed=3, use_graph=True, precision="fp16")
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 4096, 58))
    p = d.update_plan()
    print("fp16", B, p)
    assert {"fp16", "head_seed_fused"} <= set(p["forms"]), p
    for form in ("head_wgrad_rides_critic", "head_wgrad_rides_actor", "q_train_in_dgrad", "dqda_head_bwd"):      # (>= 1024 rows: the bandwidth-tiled head kernels)
        assert (form in p["forms"]) == (B == 512), (form, p)
    assert 0 < p["launches_in_graph"] < p["launches_single"]       # the gather rides in the previous update's last launch
    assert p["launches_in_graph"] <= FP16_LAUNCHES[B], p
    d.update_async_n(17)
    assert np.isfinite(d.read_stats()[0])
    d.close()


# kernels per update inside a sixteen-update graph (a regression bound: fewer is fine, more is a schedule that fell back)
FP16_LAUNCHES = {512: 28, 4096: 32}


def test_tuning_bits_show_in_the_plan(pkg, gpu):
    """Every A/B bit of cfg.tuning_flags that selects a separate-launch form removes its form from the plan and adds launches."""
    base = _make(pkg, dict(state_size=58, minibatch=256, hidden=TOWER), None)
    p0 = base.update_plan(); base.close()
    T = pkg.capi
    for bit, form in ((T.TUNE_SEPARATE_HEAD_SEED, "head_seed_fused"), (T.TUNE_BWD_UNSHIFTED, "bwd_shifted_critic"),
                      (T.TUNE_SEPARATE_ACTOR_HEAD_BWD, "dqda_head_bwd"), (T.TUNE_SEPARATE_Q_TRAIN, "q_train_in_dgrad"),
                      (T.TUNE_SEPARATE_FIRST_LAYER, "critic_l0_rides"), (T.TUNE_SEPARATE_CRITIC_FIRST_LAYERS, "first_layers_merged"),
                      (T.TUNE_LATE_GATHER, "early_gather_l0")):
        d = _make(pkg, dict(state_size=58, minibatch=256, hidden=TOWER, tuning=bit), None)
        p = d.update_plan(); d.close()
        assert form not in p["forms"], (bit, p)
        assert p["launches_in_graph"] >= p0["launches_in_graph"] and p["launches_single"] >= p0["launches_single"], (bit, p, p0)
        if bit != T.TUNE_BWD_UNSHIFTED:                         # (the unshifted schedule has the same launch count)
            assert p["launches_in_graph"] > p0["launches_in_graph"] or p["launches_single"] > p0["launches_single"], (bit, p, p0)


def test_plan_refused_mid_update(pkg, gpu):
    d = _make(pkg, dict(state_size=58, minibatch=32, hidden=(64, 64)), None)
    d.update_phase(0)
    with pytest.raises(pkg.DQNFatal, match="phased update"):
        d.update_plan()
    d.update_phase(1); d.update_phase(2)
    assert d.update_plan()["launches_single"] > 0
    d.close()
