"""dqnhip_config.tuning_flags: every bit selects an alternative SCHEDULE of the same arithmetic (the library
reads no environment variable).  Each bit is run against the default on identical inputs in one process."""
import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu


def _run(pkg, tuning, B, hidden, n_up=3, **kw):
    d = pkg.DQN(58, minibatch=B, hidden=hidden, memory=4096, seed=3, precision="fp16", tuning=tuning, **kw)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 2000, 58))
    rng = np.random.default_rng(7)
    stats, grads = [], []
    for _ in range(n_up):
        idx = rng.integers(0, 2000, B).astype(np.int32)
        d.update_phase(0, idx); gc = d.get_params(1, pkg.KIND_G)
        d.update_phase(1); ga = d.get_params(0, pkg.KIND_G)
        d.update_phase(2)
        stats.append(d.read_stats()); grads.append((ga, gc))
    w = [d.get_params(n).astype(np.float64) for n in range(4)]
    d.close()
    return stats, grads, w


@pytest.mark.parametrize("B,hidden", [
    (256, (256, 128, 128)),                    # below kGroupMinRows: both run the per-layer form (bit-identical)
    (512, (256, 128, 128)),                    # grouped wgrads on 128x128 tiles against dgrad + wgrad pairs on 64x64 split-K
    (1024, (256, 256)),
    (2048, (1024, 1024, 1024, 1024)),          # the BASELINE tower: 3 x 64 + 8 big tiles + 64 column-sum workgroups
    (128, (128, 128, 128, 128, 128)),          # 5 layers > 4 problems per launch: falls back to the per-layer form
])
def test_fp16_grouped_wgrad_equals_per_layer(pkg, gpu, B, hidden):
    """All wgrads of a net in ONE launch (default) against one launch per layer (DQNHIP_TUNE_FP16_WGRAD_PER_LAYER):
    same fp16 operands, same rounding points; the tiles differ (128x128 vs 64x64 split-K), i.e. only the order in
    which fp32 partial sums are added -> gradients agree to fp32 round-off, parameters after three updates too."""
    a = _run(pkg, 0, B, hidden)
    b = _run(pkg, pkg.capi.TUNE_FP16_WGRAD_PER_LAYER, B, hidden)
    assert np.allclose(a[0], b[0], rtol=1e-4, atol=1e-6), (a[0], b[0])
    for it, ((ga, gc), (gb_a, gb_c)) in enumerate(zip(a[1], b[1])):
        for x, y in ((ga, gb_a), (gc, gb_c)):
            # first update (identical weights): one rows-long fp32 chain per element against four rows/4-long ones, on
            # heavily cancelling sums -> a few 1e-5 of the gradient's norm (the fp16 rounding of the operands, common to
            # both, is ~1e-3).  Later updates start from weights that already differ by Adam steps of magnitude lr.
            rel = np.linalg.norm(x.astype(np.float64) - y) / max(np.linalg.norm(y), 1e-30)
            assert rel <= (2e-4 if it == 0 else 5e-3), (it, rel)
    # parameters: Adam's first steps have magnitude lr whatever |g| is, so an element whose gradient is ~0 may step the
    # other way: bounded by the steps taken, tiny on average
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net, (wa, wb) in enumerate(zip(a[2], b[2])):
        dd = np.abs(wa - wb)
        assert dd.max() <= 2 * 3 * lr[net] + 1e-7 and dd.mean() <= 0.01 * lr[net] + 1e-9, (net, dd.max(), dd.mean())


def test_fp16_grouped_wgrad_under_graph_replay(pkg, gpu):
    a = _run(pkg, 0, 256, (256, 128, 128))
    d = pkg.DQN(58, minibatch=256, hidden=(256, 128, 128), memory=4096, seed=3, precision="fp16", use_graph=True)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 2000, 58))
    rng = np.random.default_rng(7)
    for _ in range(3):
        d.UpdateActorCritic(rng.integers(0, 2000, 256).astype(np.int32))
    for n in range(4):
        np.testing.assert_array_equal(d.get_params(n).astype(np.float64), a[2][n])     # replayed = eager, bit for bit
    d.close()


def _run32(pkg, tuning, B, hidden, S=58, n_up=3, use_graph=False):
    d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=4096, seed=3, tuning=tuning, use_graph=use_graph)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 2000, S))
    rng = np.random.default_rng(7)
    stats, dbg = [], []
    for _ in range(n_up):
        idx = rng.integers(0, 2000, B).astype(np.int32)
        stats.append(d.UpdateActorCritic(idx))
        dbg.append((d.debug_read("q_policy"), d.debug_read("dq_da")))
    w = [d.get_params(n) for n in range(4)] + [d.get_params(n, k) for n in (0, 1) for k in (pkg.KIND_M, pkg.KIND_V, pkg.KIND_G)]
    d.close()
    return stats, dbg, w


@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]: seed from gemm_fwd_lds<2,2,true,2>'s epilogue
    (32, (1024, 512, 256, 128), 59),           # the reference's defaults: top layer K = 256 -> the direct forward kernel
    (64, (256, 128, 64, 64), 59),              # narrow top layer (gemm_fwd_direct)
    (512, (1024, 1024), 68),                   # 64x32 forward tiles
    (1024, (256, 256), 58),                    # rows >= 1024: the separate form takes k_head_bwd_big
    (128, (512,), 58),                         # one tower layer: it carries the seed AND its dgrad carries the q rider
])
def test_fused_head_seed_equals_separate_launch(pkg, gpu, B, hidden, S):
    """fp32 learner, critic(s, mu(s)) pass: the dq = -1 seed from the top layer's forward epilogue + q(s, mu(s)) as rider
    blocks of the narrow dgrad launch (default) against the head-backward launch of their own
    (DQNHIP_TUNE_SEPARATE_HEAD_SEED).  Same arithmetic on the same values: every result bit-identical."""
    a = _run32(pkg, 0, B, hidden, S)
    b = _run32(pkg, pkg.capi.TUNE_SEPARATE_HEAD_SEED, B, hidden, S)
    assert a[0] == b[0], (a[0], b[0])
    for (qa, da), (qb, db) in zip(a[1], b[1]):
        np.testing.assert_array_equal(qa, qb); np.testing.assert_array_equal(da, db)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)


def test_fused_head_seed_under_graph_replay(pkg, gpu):
    a = _run32(pkg, 0, 256, (1024, 1024, 1024, 1024), n_up=4)
    g = _run32(pkg, 0, 256, (1024, 1024, 1024, 1024), n_up=4, use_graph=True)
    assert a[0] == g[0]
    for x, y in zip(a[2], g[2]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]
    (512, (1024, 1024), 68),                   # two layers: dgrad(1) | wgrad(1) + wgrad(0)
    (256, (1024, 512, 256, 128), 59),          # the reference's tower: layers of different widths share a launch
    (1024, (256, 256), 58),                    # rows >= 1024 (no head rider: k_head_bwd_big + k_head_wred)
    (32, (1024, 512, 256, 128), 59),           # the reference's defaults: side-by-side pair launches -> not shifted (identical)
])
def test_shifted_backward_schedule_equals_per_layer_launches(pkg, gpu, B, hidden, S):
    """fp32 learner: dgrad(L-1) | wgrad(i+1) + dgrad(i) | ... | wgrad(1) + wgrad(0) + head riders (default) against
    wgrad(i) + dgrad(i) per layer + a last launch with the first layer's wgrad alone (DQNHIP_TUNE_BWD_UNSHIFTED): the same
    workgroups doing the same arithmetic in different launches — every result bit-identical.  (Both sides with k_head_q_train
    in a launch of its own: k_dgrad_qtrain exists in the shifted schedule only and has its own test below.)"""
    sep = pkg.capi.TUNE_SEPARATE_Q_TRAIN
    a = _run32(pkg, sep, B, hidden, S)
    b = _run32(pkg, sep | pkg.capi.TUNE_BWD_UNSHIFTED, B, hidden, S)
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,hidden", [
    (128, (256, 128, 128)),                    # k_head_bwd<10> carries the q rider
    (512, (1024, 1024, 1024, 1024)),           # the rank shape of BASELINE configs[4] on 8 GPUs
    (1024, (256, 256)),                        # rows >= 1024: k_head_bwd_big<10> carries it
    (2048, (1024, 1024, 1024, 1024)),          # 256x128 / 128x128 forward tiles emit the seed
])
def test_fp16_fused_head_seed_equals_separate_launch(pkg, gpu, B, hidden):
    """fp16 learner, critic(s, mu(s)) pass: the scaled fp16 seed panel from the top layer's forward epilogue (HGemm::seed_w) +
    q(s, mu(s)) as rider blocks of the actor heads' backward launch (default) against the dq = -1 head-backward launch of
    their own (DQNHIP_TUNE_SEPARATE_HEAD_SEED): the same arithmetic on the same fp16-rounded activations, bit-identical."""
    a = _run(pkg, 0, B, hidden)
    b = _run(pkg, pkg.capi.TUNE_SEPARATE_HEAD_SEED, B, hidden)
    assert a[0] == b[0], (a[0], b[0])
    for (ga, gc), (gb_a, gb_c) in zip(a[1], b[1]):
        np.testing.assert_array_equal(ga, gb_a); np.testing.assert_array_equal(gc, gb_c)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)



@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]: 16 row tiles x 4 column chunks + 64 q-rider blocks
    (32, (1024, 512, 256, 256), 59),           # 2 row tiles, one column chunk
    (64, (256, 128, 64, 64), 59),              # H = 64: a quarter of the workgroup's 256 column threads live (round 6: any tower-top width)
    (32, (1024, 512, 256, 128), 59),           # the reference's compile-time defaults (src/dqn.hpp:19, src/dqn.cpp:425): H = 128
    (512, (1024, 1024), 68),                   # 1v1 state size: the action columns start at 68 (panel 128 wide)
    (128, (512,), 58),                         # one tower layer: the fused launch reads the seed the forward epilogue left
    (96, (256, 256, 256), 77),                 # 2v1 state size, a row count that is not a multiple of 64
])
def test_fused_dqda_and_actor_head_backward_equals_separate_launches(pkg, gpu, B, hidden, S):
    """fp32 learner: the critic's first-layer action-column input gradient, the inverting gradients and the actor heads'
    backward in ONE launch (k_dqda_head_bwd, default) against the narrow dgrad launch + k_head_bwd<10>
    (DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD).  The fused workgroups recompute the same 16 x 16 tile values and run the same head
    loop: every result bit-identical, eager and graph-replayed."""
    a = _run32(pkg, 0, B, hidden, S)
    b = _run32(pkg, pkg.capi.TUNE_SEPARATE_ACTOR_HEAD_BWD, B, hidden, S)
    assert a[0] == b[0], (a[0], b[0])
    for (qa, da), (qb, db) in zip(a[1], b[1]):
        np.testing.assert_array_equal(qa, qb); np.testing.assert_array_equal(da, db)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    g = _run32(pkg, 0, B, hidden, S, use_graph=True)
    assert a[0] == g[0]
    for x, y in zip(a[2], g[2]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]
    (512, (1024, 1024), 68),                   # two layers: the fused launch is the whole dgrad chain of Step(1)
    (256, (512, 1024, 512), 59),               # layers of different widths: 8 column tiles write 8 slices of a 512-wide dZ_L
    (64, (1024, 512), 59),                     # 32 tiles of 64 x 16 at 64 rows
    (32, (1024, 512, 256, 128), 59),           # the reference's defaults: pair launches, top layer 128 wide -> both sides separate (identical)
])
def test_q_train_inside_the_top_dgrad_equals_its_own_launch(pkg, gpu, B, hidden, S):
    """fp32 learner, Step(1): q', q, the TD target, the loss, dq and the tower-top gradient inside the critic's top-layer dgrad
    launch (k_dgrad_qtrain, default: the head dot products arrive in 16-column pieces from the top forward layers' epilogues)
    against k_head_q_train in a launch of its own (DQNHIP_TUNE_SEPARATE_Q_TRAIN).  The dot product is summed in another fixed
    order and the top layer's input gradient takes dq_r after the reduction instead of before: fp32 round-off, so the per-row
    outputs agree to a few ulp of the row's |x||w|, the gradients to ~1e-6 of their norm and the parameters to the usual
    Adam-step bound."""
    out = {}
    for tuning in (0, pkg.capi.TUNE_SEPARATE_Q_TRAIN):
        d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=4096, seed=3, tuning=tuning)
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 2000, S))
        rng = np.random.default_rng(7)
        rec = []
        for u in range(3):
            idx = rng.integers(0, 2000, B).astype(np.int32)
            d.update_phase(0, idx)
            first = {k: d.debug_read(k) for k in ("q_target", "q_train", "y")}
            first["gc"] = d.get_params(1, pkg.KIND_G)
            d.update_phase(1); d.update_phase(2)
            first["stats"] = d.read_stats()
            rec.append(first)
        out[tuning] = (rec, [d.get_params(n).astype(np.float64) for n in range(4)])
        d.close()
    (ra, wa), (rb, wb) = out[0], out[pkg.capi.TUNE_SEPARATE_Q_TRAIN]
    for k in ("q_target", "q_train", "y"):                           # first update: same weights
        scale = max(1.0, float(np.abs(rb[0][k]).max()))
        np.testing.assert_allclose(ra[0][k], rb[0][k], rtol=0, atol=4e-6 * scale)
    assert abs(ra[0]["stats"][0] - rb[0]["stats"][0]) <= 1e-5 * max(1.0, abs(rb[0]["stats"][0]))     # the reported (pre-update) loss
    rel = np.linalg.norm(ra[0]["gc"].astype(np.float64) - rb[0]["gc"]) / np.linalg.norm(rb[0]["gc"])
    assert rel <= 2e-5, rel
    for a, b in zip(ra, rb):
        assert np.allclose(a["stats"], b["stats"], rtol=1e-4, atol=1e-6), (a["stats"], b["stats"])
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net, (x, y) in enumerate(zip(wa, wb)):
        dd = np.abs(x - y)
        assert dd.max() <= 2 * 3 * lr[net] + 1e-7 and dd.mean() <= 0.01 * lr[net] + 1e-9, (net, dd.max(), dd.mean())


@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]: K = 68 -> 128, two float4 of W1 per rider thread
    (256, (1024, 1024), 40),                   # K = 50 -> 64: one float4 per thread
    (512, (1024, 512, 256, 128), 59),          # the reference's tower at 512 rows
    (32, (1024, 512, 256, 128), 59),           # the reference's defaults: two row tiles per rider
    (64, (192, 64), 77),                       # 12 riders; K = 87 -> 128
    (1024, (256, 256), 58),                    # more than 512 rows: the layer keeps its own launch on both sides
])
def test_first_layer_inside_the_optimiser_launch_equals_its_own_launch(pkg, gpu, B, hidden, S):
    """fp32 learner: the first tower layer of critic(s, mu(s)) computed by the optimiser workgroups that own W1 / b1, on the
    weights they have just stepped (FirstLayerRider, default), against a launch of its own (DQNHIP_TUNE_SEPARATE_FIRST_LAYER).
    The same Adam arithmetic on every element and fwd_direct_body's reduction order: every result bit-identical, eager and
    graph replay."""
    a = _run32(pkg, 0, B, hidden, S)
    b = _run32(pkg, pkg.capi.TUNE_SEPARATE_FIRST_LAYER, B, hidden, S)
    assert a[0] == b[0], (a[0], b[0])
    for (qa, da), (qb, db) in zip(a[1], b[1]):
        np.testing.assert_array_equal(qa, qb); np.testing.assert_array_equal(da, db)
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    g = _run32(pkg, 0, B, hidden, S, use_graph=True)
    assert a[0] == g[0]
    for x, y in zip(a[2], g[2]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,hidden,S", [
    (256, (1024, 1024, 1024, 1024), 58),       # BASELINE configs[1]: state half K = 64 of a 128-wide critic panel
    (256, (1024, 512, 256, 128), 59),          # the reference's tower
    (64, (256, 128), 68),                      # S = 68: the state half spans both 64-column halves of the panel (action columns still zero)
    (32, (1024, 512, 256, 128), 59),           # the reference's defaults
    (512, (192, 64), 77),                      # 192 outputs: a partial last thread group in the head kernel
    (1024, (256, 256), 58),                    # 1024 rows: the wave-per-row head kernel -> both sides keep the launch (identical)
])
def test_critic_first_layers_merged_equal_their_own_launch(pkg, gpu, B, hidden, S):
    """fp32 learner, Step(1): critic(s, a)'s first layer in the update's first GEMM launch and critic_target(s', mu'(s'))'s split into
    a state half (same launch) and a rank-10 action half applied by the target actor's head kernel (default) against the critics'
    first-layer launch behind the heads (DQNHIP_TUNE_SEPARATE_CRITIC_FIRST_LAYERS).  Only critic_target's first layer changes its
    summation order: q' and everything downstream agree to fp32 round-off."""
    out = {}
    for tuning in (0, pkg.capi.TUNE_SEPARATE_CRITIC_FIRST_LAYERS):
        d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=4096, seed=3, tuning=tuning)
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 2000, S))
        rng = np.random.default_rng(7)
        rec = []
        for u in range(3):
            idx = rng.integers(0, 2000, B).astype(np.int32)
            d.update_phase(0, idx)
            first = {k: d.debug_read(k) for k in ("q_target", "q_train", "y")}
            first["l1_ct"], first["l1_c"], first["l1_a"] = d.debug_read("act2_1"), d.debug_read("act3_1"), d.debug_read("act1_1")
            first["gc"] = d.get_params(1, pkg.KIND_G)
            d.update_phase(1); d.update_phase(2)
            first["stats"] = d.read_stats()
            rec.append(first)
        out[tuning] = (rec, [d.get_params(n).astype(np.float64) for n in range(4)])
        d.close()
    (ra, wa), (rb, wb) = out[0], out[pkg.capi.TUNE_SEPARATE_CRITIC_FIRST_LAYERS]
    # first update (same weights): the layers whose arithmetic did not change are bit-identical, critic_target's agrees to round-off
    np.testing.assert_array_equal(ra[0]["l1_c"], rb[0]["l1_c"]); np.testing.assert_array_equal(ra[0]["l1_a"], rb[0]["l1_a"])
    np.testing.assert_array_equal(ra[0]["q_train"], rb[0]["q_train"])
    scale = max(1.0, float(np.abs(rb[0]["l1_ct"]).max()))
    np.testing.assert_allclose(ra[0]["l1_ct"], rb[0]["l1_ct"], rtol=0, atol=4e-6 * scale)
    for k in ("q_target", "y"):
        sc = max(1.0, float(np.abs(rb[0][k]).max()))
        np.testing.assert_allclose(ra[0][k], rb[0][k], rtol=0, atol=4e-6 * sc)
    rel = np.linalg.norm(ra[0]["gc"].astype(np.float64) - rb[0]["gc"]) / np.linalg.norm(rb[0]["gc"])
    assert rel <= 2e-5, rel
    for a, b in zip(ra, rb):
        assert np.allclose(a["stats"], b["stats"], rtol=1e-4, atol=1e-6), (a["stats"], b["stats"])
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net, (x, y) in enumerate(zip(wa, wb)):
        dd = np.abs(x - y)
        assert dd.max() <= 2 * 3 * lr[net] + 1e-7 and dd.mean() <= 0.01 * lr[net] + 1e-9, (net, dd.max(), dd.mean())


@pytest.mark.parametrize("B,hidden,S,n", [
    (256, (1024, 1024, 1024, 1024), 58, 35),   # BASELINE configs[1]: two full graphs + three single updates
    (64, (256, 128, 64, 64), 59, 16),          # the reference-like small tower, one graph
    (32, (1024, 512, 256, 128), 59, 20),       # the reference's defaults
    (256, (1024, 1024), 68, 17),               # S = 68: the actor's panel is 128 wide -> both sides gather late (identical)
])
def test_early_gather_and_next_update_first_layers_equal_the_late_form(pkg, gpu, B, hidden, S, n):
    """fp32 learner, dqnhip_update_async_n: the next update's gather in the critic's optimiser launch and its four first layers as
    riders of the actor's optimiser launch (k_adam_soft_fwd1_gather / k_adam_soft_l0, default) against the gather in the update's
    last launch and the first layers in a launch of their own (DQNHIP_TUNE_LATE_GATHER).  The same arithmetic on the same
    values, two minibatch panels by update parity: every parameter, Adam moment and statistic bit-identical."""
    out = []
    for tuning in (0, pkg.capi.TUNE_LATE_GATHER):
        d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=4096, seed=11, tuning=tuning, use_graph=True)
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 3000, S))
        d.update_async_n(n)
        st = d.read_stats()
        w = [d.get_params(k) for k in range(4)] + [d.get_params(k, kind) for k in (0, 1) for kind in (pkg.KIND_M, pkg.KIND_V)]
        out.append((st, w, (d.actor_iter(), d.critic_iter())))
        d.close()
    assert out[0][0] == out[1][0] and out[0][2] == out[1][2] == (n, n)
    for x, y in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,hidden", [(128, (256, 128, 128)), (512, (1024, 1024, 1024, 1024)), (256, (256, 256))])
def test_fp16_fused_dqda_head_bwd_against_separate_launches(pkg, gpu, B, hidden):
    """fp16 learner (round 6): the critic's layer-0 input gradient (ten action columns), the inverting gradients and the actor heads'
    backward in ONE launch (k_dqda_head_bwd<true>) against the fp16-MFMA layer-0 dgrad launch + k_head_bwd<10>
    (DQNHIP_TUNE_SEPARATE_ACTOR_HEAD_BWD).  Same fp16 operands, exact products; the fp32 additions of the 16-column tile run in
    another order (16x16x4 fp32 MFMA chains on converted operands against the 32x32x16 fp16 MFMA), so the two agree to fp32
    round-off before the tower-top gradient is rounded to fp16 — not bit for bit."""
    a = _run(pkg, 0, B, hidden)
    b = _run(pkg, pkg.capi.TUNE_SEPARATE_ACTOR_HEAD_BWD, B, hidden)
    assert np.allclose(a[0], b[0], rtol=2e-4, atol=2e-6), (a[0], b[0])
    for it, ((ga, gc), (gb_a, gb_c)) in enumerate(zip(a[1], b[1])):
        for x, y in ((ga, gb_a), (gc, gb_c)):
            rel = np.linalg.norm(x.astype(np.float64) - y) / max(np.linalg.norm(y), 1e-30)
            assert rel <= (1e-3 if it == 0 else 5e-3), (it, rel)
    for tuning, present in ((0, True), (pkg.capi.TUNE_SEPARATE_ACTOR_HEAD_BWD, False)):
        p = pkg.DQN(58, minibatch=B, hidden=hidden, memory=4096, seed=3, precision="fp16", tuning=tuning)
        assert ("dqda_head_bwd" in p.update_plan()["forms"]) == present, (tuning, p.update_plan())
        p.close()
