"""One rank of a data-parallel group against the plain learner, same box, same process: sixteen-update graphs (dqnhip_update_async_n /
dqnhip_dp_update_n) and one-update graphs.  With ONE rank every collective is a copy, so the difference is what the data-parallel
launch sequence itself costs (k_tails, k_sumsq, the bf16 image) — VERDICT r5 item 1: within 3 % of the plain update.
usage: python scripts/dp_rank_timing.py [fp32|fp16] [rows ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from __graft_entry__ import load_package
from synth import synth_replay

pkg = load_package()
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
rows = [int(x) for x in sys.argv[2:]] or [256, 512]
HID = (1024,) * 4


def make(B, dp):
    d = pkg.DQN(58, minibatch=B, hidden=HID, memory=65536, seed=1, use_graph=True, precision=prec)
    d.add_transitions_arrays(*synth_replay(np.random.default_rng(3), 60000, 58))
    if dp is not None:
        d.dp_init(pkg.DQN.dp_unique_id(), **dp)
    return d


def rate(d, dp, n_form, n=1600):
    step_n = (d.dp_update_n if dp is not None else d.update_async_n) if n_form else None
    def run(k):
        if n_form:
            step_n(k)
        else:
            for _ in range(k):
                d.dp_update(None) if dp is not None else d.update_async(None)
    run(160); d.read_stats()
    t = time.perf_counter(); run(n); d.read_stats()
    return (time.perf_counter() - t) / n * 1e3


for B in rows:
    res = {}
    forms = [("plain", None), ("dp_fp32_allreduce", dict()), ("dp_bf16_exchange", dict(half_grads=True))]
    ds = [(name, dp, make(B, dp)) for name, dp in forms]
    for rep in range(2):
        for name, dp, d in ds:
            for n_form in (True, False):
                res.setdefault((name, n_form), []).append(rate(d, dp, n_form))
    for name, dp, d in ds:
        print(prec, B, name, d.update_plan())
        d.close()
    base = {nf: min(res[("plain", nf)]) for nf in (True, False)}
    for (name, nf), v in sorted(res.items()):
        print("%s rows=%d %-20s %-18s ms/update %s  (%.1f %% over plain)" % (prec, B, name, "16-update graphs" if nf else "1-update graphs",
              " ".join("%.4f" % x for x in v), (min(v) / base[nf] - 1) * 100))
