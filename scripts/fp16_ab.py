"""fp16 learner, ms/update by minibatch under tuning flags (A/B inside one process, alternating):
   python scripts/fp16_ab.py [flagsA flagsB] [minibatches ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from __graft_entry__ import load_package
from synth import synth_replay
pkg = load_package()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
fa, fb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
sizes = [int(x) for x in sys.argv[3:]] or [256, 512, 1024, 2048, 4096]
prec = os.environ.get("PREC", "fp16")
S, H = 58, (1024, 1024, 1024, 1024)
data = synth_replay(np.random.default_rng(7), 100000, S)
for B in sizes:
    ds = {f: pkg.DQN(S, minibatch=B, hidden=H, memory=200000, seed=1, use_graph=True, precision=prec, tuning=f) for f in {fa, fb}}
    for d in ds.values():
        d.add_transitions_arrays(*data)
        for _ in range(30): d.update_async(None)
        d.read_stats()
    res = {f: [] for f in ds}
    for rep in range(3):
        for f, d in ds.items():
            n = 200
            t0 = time.perf_counter()
            for _ in range(n): d.update_async(None)
            d.read_stats()
            res[f].append((time.perf_counter() - t0) / n * 1e3)
    print("B=%d %s" % (B, prec), {f: ["%.4f" % x for x in v] for f, v in res.items()}, flush=True)
    for d in ds.values(): d.close()
