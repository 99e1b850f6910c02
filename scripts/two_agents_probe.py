"""Two (or more) independent learners on ONE GPU, each on its own stream with its own captured update
(the reference runs one DQN per agent thread against one GPU, src/dqn_main.cpp:62, 264): aggregate updates/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from synth import synth_replay
import ctypes
pkg = load_package()
hip = ctypes.CDLL("libamdhip64.so")
S, HID = 58, (1024,) * 4
for n_agents in (1, 2, 3, 4):
    ds = [pkg.DQN(S, minibatch=256, hidden=HID, memory=100000, seed=1 + i, use_graph=True) for i in range(n_agents)]
    rng = np.random.default_rng(1)
    for d in ds:
        d.add_transitions_arrays(*synth_replay(rng, 50000, S))
    for _ in range(50):
        for d in ds: d.update_async(None)
    hip.hipDeviceSynchronize()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n):
        for d in ds: d.update_async(None)
    hip.hipDeviceSynchronize()
    dt = time.perf_counter() - t0
    print("agents=%d  aggregate %.0f updates/s  (%.1f per agent, %.3f ms per round of %d)" % (n_agents, n * n_agents / dt, n / dt, dt / n * 1e3, n_agents), flush=True)
    for d in ds: d.close()
