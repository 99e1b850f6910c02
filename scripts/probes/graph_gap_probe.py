"""How much of an update's wall clock is the gap between two hipGraph launches?  (inside gpurun)
   single: one update per hipGraphLaunch (dqnhip_update_async); multi: dqnhip_update_async_n (sixteen updates per launch,
   the gather of update u + 1 riding in update u's last launch).  (A third mode — two instances of the single-update graph
   launched alternately — measured the same as single and was removed: profiles/r04_graph_gap.txt.)"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from __graft_entry__ import load_package
sys.path.insert(0, "/root/repo/tests")
from synth import synth_replay
pkg = load_package()

def run(mode, n=4000):
    d = pkg.DQN(58, minibatch=256, hidden=(1024,) * 4, memory=100000, seed=1, use_graph=True)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(5), 50000, 58))
    step = (lambda k: d.update_async_n(k)) if mode == "multi" else (lambda k: [d.update_async(None) for _ in range(k)])
    step(800); d.read_stats()
    t = time.perf_counter(); step(n); d.read_stats(); dt = time.perf_counter() - t
    w = [d.get_params(i).copy() for i in range(4)]
    d.close()
    return dt / n * 1e3, w

for rep in range(3):
    r = {m: run(m) for m in ("single", "multi")}
    print("ms/update: " + "  ".join("%s %.5f" % (m, r[m][0]) for m in r),
          "| multi == single bitwise:", all(np.array_equal(a, b) for a, b in zip(r["single"][1], r["multi"][1])), flush=True)
