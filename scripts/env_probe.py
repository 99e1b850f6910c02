"""env-steps/s of the batched front-end at BASELINE.json configs[2] (1v1: S = 68, 64 workers) and wider."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 68
WORKERS = [int(x) for x in sys.argv[2:]] or [64, 256, 2048]
for workers in WORKERS:
    d = pkg.DQN(S, minibatch=256, hidden=(1024,) * 4, memory=1200000, seed=1, use_graph=True)
    env = pkg.EnvFrontEnd(d, workers, max_steps=500, p_end=0.01, seed=5)
    env.step(0.1, 40); env.stats()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); env.step(0.1, 320); env.stats(); best = min(best, (time.perf_counter() - t) / 320)
    print("S=%d workers=%4d  %.2f us per batched step  %.3f M env-steps/s" % (S, workers, best * 1e6, workers / best / 1e6), flush=True)
    env.close(); d.close()
