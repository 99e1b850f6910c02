"""Long-run check of dqnhip_update_async_n (sixteen updates per hipGraph launch, gathers riding ahead) against single calls:
two learners, same seed, N updates each in uneven bursts, new transitions between bursts — bit-identical at the end.  (inside gpurun)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from __graft_entry__ import load_package
from synth import synth_replay
pkg = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B, hidden = (256, (1024,) * 4) if prec == "fp32" else (512, (1024,) * 4)
ds = [pkg.DQN(58, minibatch=B, hidden=hidden, memory=200000, seed=5, use_graph=True, precision=prec) for _ in range(2)]
first = synth_replay(np.random.default_rng(1), 50000, 58)
for d in ds:
    d.add_transitions_arrays(*first)
rng = np.random.default_rng(7)
done = 0; t0 = time.time(); burst_no = 0
while done < N:
    k = int(rng.integers(1, 700))
    k = min(k, N - done)
    for _ in range(k):
        ds[0].update_async(None)
    ds[1].update_async_n(k)
    done += k; burst_no += 1
    if burst_no % 5 == 0:
        more = synth_replay(np.random.default_rng(100 + burst_no), int(rng.integers(50, 3000)), 58)
        for d in ds:
            d.add_transitions_arrays(*more)
    if burst_no % 20 == 0:
        sa, sb = ds[0].read_stats(), ds[1].read_stats()
        assert sa == sb and all(np.isfinite(sa)), (done, sa, sb)
sa, sb = ds[0].read_stats(), ds[1].read_stats()
same = sa == sb and all(np.array_equal(ds[0].get_params(n), ds[1].get_params(n)) for n in range(4)) and \
    all(np.array_equal(ds[0].get_params(n, k), ds[1].get_params(n, k)) for n in (0, 1) for k in (pkg.KIND_M, pkg.KIND_V))
print("%s: %d updates in %d bursts, %.1f s; stats %s; iters %s / %s; bit-identical: %s" % (
    prec, done, burst_no, time.time() - t0, sa, (ds[0].actor_iter(), ds[0].critic_iter()), (ds[1].actor_iter(), ds[1].critic_iter()), same))
assert same
