"""64 env workers as ONE batched front-end against TWO groups of 32 on two streams (VERDICT r3 "Next" item 7).
Upper bound of the idea: the two groups here are two independent learners (own weights, own ring, own stream) — a real
implementation would share both and add cross-stream ordering on the ring.   python scripts/env_two_groups_probe.py [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 68
H = (1024, 1024, 1024, 1024)
def mk(workers, seed):
    d = pkg.DQN(S, minibatch=256, hidden=H, memory=200000, seed=seed, use_graph=True)
    e = pkg.EnvFrontEnd(d, workers, max_steps=500, p_end=0.01, seed=5 + seed)
    e.step(0.1, 32); e.stats()
    return d, e
def run(envs, n):
    for e in envs: e.stats()
    t = time.perf_counter()
    for _ in range(n // 16):
        for e in envs: e.step(0.1, 16)          # asynchronous: 16-step graph replays, alternating streams
    for e in envs: e.stats()
    return (time.perf_counter() - t) / n
for rep in range(3):
    d, e = mk(64, 1)
    dt = run([e], 1600)
    print("one front-end, 64 workers      : %6.2f us per batched step  %.2f M env-steps/s" % (dt * 1e6, 64 / dt / 1e6), flush=True)
    e.close(); d.close()
    (d1, e1), (d2, e2) = mk(32, 1), mk(32, 2)
    dt = run([e1, e2], 1600)
    print("two front-ends x 32, two streams: %6.2f us per step of both     %.2f M env-steps/s" % (dt * 1e6, 64 / dt / 1e6), flush=True)
    for x in (e1, e2, d1, d2): x.close()
    (d1, e1), (d2, e2) = mk(64, 1), mk(64, 2)
    dt = run([e1, e2], 1600)
    print("two front-ends x 64, two streams: %6.2f us per step of both     %.2f M env-steps/s (128 workers)" % (dt * 1e6, 128 / dt / 1e6), flush=True)
    for x in (e1, e2, d1, d2): x.close()
