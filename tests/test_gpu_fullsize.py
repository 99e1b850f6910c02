"""BASELINE.json's full sizes (B=256, 4x1024, 1M-transition replay): properties that do not need
the oracle to run at that size."""
import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu

B, S, HID, CAP = 256, 58, (1024, 1024, 1024, 1024), 1_000_000


@pytest.fixture(scope="module")
def big(pkg, gpu):
    dqn = pkg.DQN(S, minibatch=B, hidden=HID, memory=CAP, seed=4)
    rng = np.random.default_rng(4)
    chunks = []
    done = 0
    while done < CAP - 1:
        n = min(131072, CAP - 1 - done)
        d = synth_replay(rng, n, S)
        if done == 0 or done + n >= CAP - 1:
            chunks.append((done, d))
        dqn.add_transitions_arrays(*d)
        done += n
    yield dqn, chunks
    dqn.close()


def test_ring_holds_capacity_minus_one(big):
    dqn, chunks = big
    assert dqn.memory_size() == CAP - 1
    first_off, first = chunks[0]
    last_off, last = chunks[-1]
    for off, d in ((first_off, first), (last_off, last)):
        got = dqn.read_memory(off, 1000)
        for x, y in zip(got, d):
            np.testing.assert_array_equal(x, y[:1000])


def test_device_sampling_is_uniform_and_gathers_the_sampled_rows(big):
    from oracle import c_oracle
    dqn, _ = big
    seen = []
    u0 = dqn.actor_iter()                 # the sampling counter ticks once per update, like the iterations
    for u in range(40):
        dqn.UpdateActorCritic()
        idx = dqn.debug_read("idx").astype(np.int64)
        # the on-device sampler IS its CPU twin (oracle/dqn_oracle.c orc_philox_index): exact
        np.testing.assert_array_equal(idx, c_oracle.philox_indices(4, u0 + u, B, CAP - 1))
        assert idx.min() >= 0 and idx.max() < CAP - 1
        seen.append(idx)
        term_gathered = dqn.debug_read("terminal")
        # the gathered terminal flags are the ring's flags at the sampled logical indices
        want = np.array([dqn.read_memory(int(i), 1)[5][0] for i in idx[:8]])
        np.testing.assert_array_equal(term_gathered[:8], want)
    allidx = np.concatenate(seen)
    # 10240 uniform draws over ~1M slots: mean within 3 sigma, all quarters populated
    assert abs(allidx.mean() - (CAP - 1) / 2) < 3 * (CAP / np.sqrt(12)) / np.sqrt(allidx.size)
    assert np.all(np.histogram(allidx, bins=4, range=(0, CAP))[0] > allidx.size / 4 * 0.85)
    assert np.unique(allidx).size > 0.99 * allidx.size          # with replacement, but collisions are rare


def test_updates_are_finite_and_iterate(big):
    dqn, _ = big
    a0, c0 = dqn.actor_iter(), dqn.critic_iter()
    for _ in range(10):
        loss, q = dqn.UpdateActorCritic()
        assert np.isfinite(loss) and np.isfinite(q)
    assert (dqn.actor_iter(), dqn.critic_iter()) == (a0 + 10, c0 + 10)
    for net in range(4):
        assert np.isfinite(dqn.get_params(net)).all()
    # target nets trail the online nets by the soft update: |theta' - theta| << |theta|
    d = np.abs(dqn.get_params(3) - dqn.get_params(1)).max()
    assert 0 < d < 0.1


def test_benchmark_hook_reports_plausible_time(big):
    dqn, _ = big
    ms = dqn.Benchmark(iterations=50, warmup=5)              # DQN::Benchmark, src/dqn.cpp:487-498
    assert 0.05 < ms < 5.0
