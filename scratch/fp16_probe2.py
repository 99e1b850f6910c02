import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
np.set_printoptions(precision=4, linewidth=200, suppress=False)
shape = dict(B=128, S=59, hidden=(256, 128, 128, 128), wscale=5.0)
dqn, orc, data, rng = make_pair(pkg, n_replay=2048, precision="fp16", **shape)
idx = rng.integers(0, 2048, size=128)
dqn.update_phase(0, idx); orc.update_phase(0, idx)
dqn.update_phase(1); orc.update_phase(1, idx)
a, b = dqn.debug_read("dq_da"), orc.debug_read("dq_da")
ao, bo = dqn.debug_read("actor_out"), orc.debug_read("actor_out")
err = np.abs(a - b)
print("max |ref| per col", np.abs(b).max(0))
print("max err per col", err.max(0))
r, c = np.unravel_index(err.argmax(), err.shape)
print("worst row", r, "col", c); print(a[r]); print(b[r]); print("actor_out", ao[r]); print(bo[r])
print("rel fro per col", np.linalg.norm(a - b, axis=0) / np.linalg.norm(b, axis=0))
