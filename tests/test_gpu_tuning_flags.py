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
