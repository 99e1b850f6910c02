import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
def fro(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
for shape in [dict(B=128, S=59, hidden=(256, 128, 128, 128), wscale=5.0), dict(B=256, S=58, hidden=(1024,)*4, wscale=2.0), dict(B=128, S=59, hidden=(256, 128, 128, 128), wscale=1.0)]:
    for ug in (0, 1):
        dqn, orc, data, rng = make_pair(pkg, n_replay=2048, precision="fp16", use_graph=ug, **shape)
        B = shape["B"]
        for it in range(3):
            idx = rng.integers(0, 2048, size=B)
            if ug:
                l1, q1 = dqn.UpdateActorCritic(idx); l2, q2 = orc.update(idx)
            else:
                dqn.update_phase(0, idx); orc.update_phase(0, idx)
                gc = fro(dqn.get_params(1, 3), orc.grad_view(1))
                dqn.update_phase(1); orc.update_phase(1, idx)
                ga = fro(dqn.get_params(0, 3), orc.grad_view(0))
                dqn.update_phase(2); orc.update_phase(2, idx)
                l1, q1 = dqn.read_stats(); l2, q2 = orc.last_stats()
                print("   grads fro: critic %.2e actor %.2e" % (gc, ga))
            errs = {n: np.abs(dqn.debug_read(n) - orc.debug_read(n)).max() / max(1e-9, np.abs(orc.debug_read(n)).max()) for n in ("q_target", "y", "q_train", "q_policy", "actor_out", "dq_da")}
            print(shape["hidden"][0], "graph", ug, "it", it, "loss %.5g/%.5g q %.5g/%.5g" % (l1, l2, q1, q2), {k: "%.1e" % v for k, v in errs.items()})
        for net in range(4):
            d = np.abs(dqn.get_params(net) - orc.get_params(net))
            print("   net", net, "w max diff %.2e mean %.2e" % (d.max(), d.mean()))
        dqn.close(); orc.close()
