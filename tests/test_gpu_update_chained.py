"""dqnhip_update_chained: the drop-in's blocking update with the NEXT call's indices known one call ahead (the reference's bursts of
Update(), src/dqn_main.cpp:359-361, each drawing its indices on the host, src/dqn.cpp:501-509).  The next update's gather and first
layers ride in this update's optimiser launches; whatever rode along is used only if the prediction held and nothing changed in
between.  Every state must be exactly what plain dqnhip_update calls on the same indices leave."""
import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu


def _state(d, pkg):
    out = [d.get_params(n) for n in range(4)]
    out += [d.get_params(n, k) for n in (0, 1) for k in (pkg.KIND_M, pkg.KIND_V)]
    return out, (d.actor_iter(), d.critic_iter())


def _mk(pkg, B, hidden, S=58, **kw):
    d = pkg.DQN(S, minibatch=B, hidden=hidden, memory=4096, seed=11, use_graph=True, **kw)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 3000, S))
    return d


SHAPES = [(64, (256, 128, 64, 64), 58), (256, (1024, 1024, 1024, 1024), 58), (32, (1024, 512, 256, 128), 59),
          (64, (256, 128), 68)]     # (S = 68: the riders do not fit -> the call is dqnhip_update)


@pytest.mark.parametrize("B,hidden,S", SHAPES)
def test_chained_burst_equals_plain_updates(pkg, gpu, B, hidden, S):
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 3000, size=(23, B)).astype(np.int32)
    a = _mk(pkg, B, hidden, S)
    ra = [a.UpdateActorCritic(i) for i in idx]
    sa = _state(a, pkg); a.close()
    b = _mk(pkg, B, hidden, S)
    rb = [b.UpdateActorCriticChained(idx[t], idx[t + 1] if t + 1 < len(idx) else None) for t in range(len(idx))]
    sb = _state(b, pkg); b.close()
    assert ra == rb                                  # (critic_loss, avg_q) of every update, bit for bit
    assert sa[1] == sb[1] == (23, 23)
    for x, y in zip(sa[0], sb[0]):
        np.testing.assert_array_equal(x, y)


def test_broken_predictions_and_interleaved_entry_points(pkg, gpu):
    """a wrong prediction, new transitions, a parameter write, an asynchronous update, an env-free gap without a prediction: after each
    the next call must start a fresh chain — and the results stay those of plain updates"""
    B, hidden = 64, (256, 128, 64, 64)
    rng = np.random.default_rng(6)
    idx = rng.integers(0, 3000, size=(40, B)).astype(np.int32)
    extra = synth_replay(np.random.default_rng(9), 300, 58)
    res = []
    for chained in (False, True):
        d = _mk(pkg, B, hidden)
        up = (lambda t, nxt: d.UpdateActorCriticChained(idx[t], nxt)) if chained else (lambda t, nxt: d.UpdateActorCritic(idx[t]))
        out = []
        for t in range(5):
            out.append(up(t, idx[t + 1]))
        out.append(up(5, idx[9]))                        # predicts 9 ...
        out.append(up(6, idx[7]))                        # ... but 6 comes: fresh chain
        out.append(up(7, idx[8]))
        d.add_transitions_arrays(*extra)                 # the ring moved under the rider's gather
        out.append(up(8, idx[9]))
        w = d.get_params(1); d.set_params(1, w * 1.001)  # the first layers the riders computed are stale
        out.append(up(9, idx[10]))
        d.update_async(idx[10]); d.read_stats()          # another entry point ran the predicted update itself
        out.append(up(11, None))                         # no prediction
        out.append(up(12, idx[13]))
        d.CloneNet(0)
        for t in range(13, 30):
            out.append(up(t, idx[t + 1]))
        res.append((out, _state(d, pkg))); d.close()
    assert res[0][0] == res[1][0]
    assert res[0][1][1] == res[1][1][1]
    for x, y in zip(res[0][1][0], res[1][1][0]):
        np.testing.assert_array_equal(x, y)


def test_chained_validates_both_index_vectors(pkg, gpu):
    d = _mk(pkg, 32, (64, 64))
    ok = np.arange(32, dtype=np.int32)
    bad = ok.copy(); bad[3] = 5000
    with pytest.raises(pkg.DQNFatal, match="out of range"):
        d.UpdateActorCriticChained(bad, ok)
    with pytest.raises(pkg.DQNFatal, match="next sampled index"):
        d.UpdateActorCriticChained(ok, bad)
    d.UpdateActorCriticChained(ok, ok)
    assert d.actor_iter() == 1
    d.close()


def test_blocking_benchmark_chained_form_runs(pkg, gpu):
    d = _mk(pkg, 64, (256, 128, 64, 64))
    ms = d.BenchmarkBlocking(64, 8, seed=3, pipelined=2)
    assert ms > 0 and d.actor_iter() == 72
    d.close()
