"""The reference's compile-time default shape (minibatch 32, S = 59, tower 1024-512-256-128: src/dqn.hpp:19, src/dqn.cpp:425) on the
GPU: captured updates, for rocprofv3 --kernel-trace --stats.   usage: ref_shape_probe.py [n_updates] [minibatch] [fp32|fp16] [hidden,hidden,...]"""
import sys, os, time
import torch  # noqa: F401  (first: under rocprofv3 the system HIP runtime segfaults inside hipGraphLaunch after ~15 k traced graph
              # kernel nodes — any shape, only with the profiler attached; with torch's bundled runtime loaded first it does not)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
from synth import synth_replay
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
hid = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (1024, 512, 256, 128)
d = pkg.DQN(59, minibatch=B, hidden=hid, memory=100000, seed=1, use_graph=True, precision=prec)
d.add_transitions_arrays(*synth_replay(np.random.default_rng(3), 50000, 59))
for _ in range(100): d.update_async(None)
d.read_stats()
t0 = time.perf_counter()
for _ in range(n): d.update_async(None)
d.read_stats()
dt = (time.perf_counter() - t0) / n
print("B=%d %s %s: %.4f ms per update, %.1f updates/s" % (B, prec, hid, dt * 1e3, 1 / dt))
d.close()
