"""Soak: env front-end feeding the learner for many steps (ring wrap-around, graph replay, both precisions)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
for precision, B in (("fp32", 256), ("fp16", 256)):
    dqn = pkg.DQN(58, minibatch=B, hidden=(1024,) * 4, memory=60000, seed=3, use_graph=True, precision=precision)
    env = pkg.EnvFrontEnd(dqn, 64, max_steps=500, p_end=0.01, seed=9)
    t0 = time.time(); n_upd = 0
    for it in range(400):
        env.step(max(0.1, 1.0 - it / 200.0), 50)            # 3200 env steps
        if dqn.memory_size() >= 1000:
            for _ in range(20):
                dqn.update_async(None); n_upd += 1
            loss, q = dqn.read_stats()
            assert np.isfinite(loss) and np.isfinite(q), (it, loss, q)
    steps, eps, rsum, goals = env.stats()
    print(precision, "env steps", steps, "episodes", eps, "updates", n_upd, "memory", dqn.memory_size(),
          "loss %.4g avg_q %.4g" % (loss, q), "iters", dqn.actor_iter(), "%.1fs" % (time.time() - t0), flush=True)
    assert dqn.memory_size() == 59999 and dqn.actor_iter() == n_upd
    w = dqn.get_params(0); assert np.isfinite(w).all()
    env.close(); dqn.close()
print("soak OK")
