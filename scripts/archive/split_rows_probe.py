"""Would splitting a 4096-row fp16 update by rows over two streams pay?  Proxy: N independent fp16 learners of 4096/N rows each,
own stream, own captured update, against ONE learner of 4096 rows — aggregate samples/s.  (Independent learners are an upper
bound for a row-split update: the split one still has to join for the weight gradients and the optimiser pass.)"""
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
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
data = synth_replay(np.random.default_rng(1), 50000, S)
for n_agents in (1, 2, 4):
    B = total // n_agents
    ds = [pkg.DQN(S, minibatch=B, hidden=HID, memory=100000, seed=1 + i, use_graph=True, precision=prec) for i in range(n_agents)]
    for d in ds:
        d.add_transitions_arrays(*data)
    for _ in range(30):
        for d in ds: d.update_async(None)
    hip.hipDeviceSynchronize()
    n = 300
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            for d in ds: d.update_async(None)
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print("%s: %d learner(s) x %d rows: %.4f ms per round of %d rows = %.2f M samples/s" % (prec, n_agents, B, best * 1e3, total, total / best / 1e6), flush=True)
    for d in ds: d.close()
