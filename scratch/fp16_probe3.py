import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
shape = dict(B=128, S=59, hidden=(256, 128, 128, 128), wscale=5.0)
res = {}
for ls in (0.0625, 1.0, 16.0):
    dqn, orc, data, rng = make_pair(pkg, n_replay=2048, precision="fp16", loss_scale=ls, **shape)
    idx = rng.integers(0, 2048, size=128)
    dqn.update_phase(0, idx); orc.update_phase(0, idx)
    gc = dqn.get_params(1, 3).copy(); gco = orc.grad_view(1).copy()
    dqn.update_phase(1); orc.update_phase(1, idx)
    res[ls] = (dqn.debug_read("dq_da").copy(), dqn.get_params(0, 3).copy(), gc)
    b = orc.debug_read("dq_da"); ga = orc.grad_view(0).copy()
    f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
    print("ls", ls, "dq_da fro vs oracle %.3e  actor grad %.3e critic grad %.3e" % (f(res[ls][0], b), f(res[ls][1], ga), f(gc, gco)))
    dqn.close(); orc.close()
f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
for ls in (0.0625, 16.0):
    print("ls", ls, "vs ls 1: dq_da %.3e actor grad %.3e critic grad %.3e" % (f(res[ls][0], res[1.0][0]), f(res[ls][1], res[1.0][1]), f(res[ls][2], res[1.0][2])))
