"""Long randomised mix of everything the C-ABI offers, for ~N seconds: env steps, updates (captured and not),
host AddTransitions, snapshots + restores into a second learner, parameter sharing on/off, replay file round
trips.  Asserts finiteness and the invariants that must hold between the two learners."""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
from synth import synth_replay
pkg = load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"        # fp16: minibatch 128; parameter sharing is an fp32-only feature
MB = 128 if prec == "fp16" else 64
S, hid = 58, (256, 128, 128)
tmp = tempfile.mkdtemp()
A = pkg.DQN(S, minibatch=MB, hidden=hid, memory=20000, seed=1, use_graph=True, save_path=os.path.join(tmp, "a"), precision=prec)
Bl = pkg.DQN(S, minibatch=MB, hidden=hid, memory=20000, seed=2, use_graph=False, save_path=os.path.join(tmp, "b"), precision=prec)
DP = pkg.DQN(S, minibatch=MB, hidden=hid, memory=20000, seed=5, use_graph=True, precision=prec, dp_world=1, dp_rank=0)
DP.dp_init(pkg.DQN.dp_unique_id(), per_layer=(prec == "fp32"), half_grads=(prec == "fp16"))
DP.add_transitions_arrays(*synth_replay(rng, 3000, S, mean_len=20))
envA = pkg.EnvFrontEnd(A, 48, max_steps=80, p_end=0.03, seed=3)
envB = pkg.EnvFrontEnd(Bl, 16, max_steps=80, p_end=0.03, seed=4)
shared_w = shared_r = False
t0 = time.time(); ops = 0; counts = {}
while time.time() - t0 < budget:
    op = rng.choice(["envA", "envB", "updA", "updB", "add", "share_w", "share_r", "snap", "file", "act", "pipe", "dp", "apply"],
                    p=[.18, .13, .17, .12, .08, .05, .02, .05, .05, .05, .04, .04, .02])
    counts[op] = counts.get(op, 0) + 1; ops += 1
    if op == "envA": envA.step(float(rng.random()), int(rng.integers(1, 40)))
    elif op == "envB": envB.step(float(rng.random()), int(rng.integers(1, 40)))
    elif op in ("updA", "updB"):
        d = A if op == "updA" else Bl
        if d.memory_size() >= 200:
            for _ in range(int(rng.integers(1, 8))):
                l, q = d.UpdateActorCritic()
                assert np.isfinite(l) and np.isfinite(q), (op, l, q)
    elif op == "pipe" and A.memory_size() >= 200:         # one-deep pipelined read-back (dqnhip_update_pipelined), then drained
        for _ in range(int(rng.integers(2, 9))):
            l, q = A.UpdateActorCriticPipelined(rng.integers(0, A.memory_size(), MB) if rng.random() < 0.5 else None)
            assert np.isfinite(l) and np.isfinite(q)
        assert all(np.isfinite(A.read_stats()))
    elif op == "dp" and DP.memory_size() >= 200:          # captured data-parallel update on a one-rank communicator (bf16 exchange for fp16)
        for _ in range(int(rng.integers(1, 6))):
            DP.dp_update(None)
        assert all(np.isfinite(DP.read_stats()))
    elif op == "apply" and A.actor_iter() > 0 and not shared_w:   # an isolated ApplyUpdate on the gradient left in the arena
        A.apply_update(int(rng.integers(0, 2)))
        assert np.isfinite(A.get_params(0)).all()
    elif op == "add":
        d = A if rng.random() < 0.5 else Bl
        d.add_transitions_arrays(*synth_replay(rng, int(rng.integers(1, 3000)), S, mean_len=20))
    elif op == "share_w" and prec == "fp32":
        shared_w = not shared_w
        A.ShareParameters(Bl, 2 if shared_w else 0, 1 if shared_w else 0)
    elif op == "share_r" and not shared_r:
        A.ShareReplayMemory(Bl); shared_r = True
    elif op == "snap" and A.actor_iter() > 0:
        A.Snapshot(os.path.join(tmp, "snapA"), False, False)
        a, c, m = pkg.FindLatestSnapshot(os.path.join(tmp, "snapA"))
        R = pkg.DQN(S, minibatch=MB, hidden=hid, memory=100, seed=9, precision=prec)
        R.RestoreActorSolver(a); R.RestoreCriticSolver(c)
        for net in range(2):
            np.testing.assert_array_equal(R.get_params(net), A.get_params(net))
        assert R.actor_iter() == A.actor_iter()
        R.close(); pkg.RemoveFilesMatchingRegexp(os.path.join(tmp, "snapA") + "_.*")
    elif op == "file" and Bl.memory_size() > 10:
        path = os.path.join(tmp, "m.replaymemory")
        Bl.SnapshotReplayMemory(path)
        R = pkg.DQN(S, minibatch=64, hidden=(64,), memory=20000)
        R.LoadReplayMemory(path)
        n = R.memory_size(); assert n == Bl.memory_size()
        x, y = R.read_memory(0, min(n, 500)), Bl.read_memory(0, min(n, 500))
        np.testing.assert_array_equal(x[0], y[0]); np.testing.assert_array_equal(x[3], y[3])
        R.close()
    elif op == "act":
        s = synth_replay(rng, 33, S)[0]
        o = A.SelectActions(s, 0.0); assert np.isfinite(o).all()
    if shared_r: assert A.memory_size() == Bl.memory_size()
    if shared_w and ops % 7 == 0:
        k = 58 * 256 + 256
        np.testing.assert_array_equal(A.get_params(0)[:k], Bl.get_params(0)[:k])
for d in (A, Bl):
    for net in range(4):
        assert np.isfinite(d.get_params(net)).all()
print("chaos OK: %d ops in %.0fs" % (ops, time.time() - t0), counts, "iters", A.actor_iter(), Bl.actor_iter(), "memory", A.memory_size(), Bl.memory_size())
envB.close(); envA.close()
if shared_w: A.ShareParameters(Bl, 0, 0)
assert DP.dp_graph_active() or DP.actor_iter() == 0
Bl.close(); A.close(); DP.close()
