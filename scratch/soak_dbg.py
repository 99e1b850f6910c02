import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
pkg = load_package()
precision = sys.argv[1]; graph = int(sys.argv[2]); mem = int(sys.argv[3])
dqn = pkg.DQN(58, minibatch=256, hidden=(1024,) * 4, memory=mem, seed=3, use_graph=bool(graph), precision=precision)
env = pkg.EnvFrontEnd(dqn, 64, max_steps=500, p_end=0.01, seed=9)
n_upd = 0
for it in range(400):
    env.step(max(0.1, 1.0 - it / 200.0), 50)
    s = env.stats()
    if dqn.memory_size() >= 1000:
        for _ in range(20):
            dqn.update_async(None); n_upd += 1
        loss, q = dqn.read_stats()
    if it % 20 == 0 or it > 360:
        print(it, "mem", dqn.memory_size(), "steps", s[0], "eps", s[1], "upd", n_upd, flush=True)
print("done", flush=True)
