"""Soak of the native data-parallel forms with a one-rank communicator: env front-end feeding the learner through ring
wrap-arounds, captured dqnhip_dp_update, every exchange form, both precisions; the sharded / replicated twins must stay
within Adam-step distance of a plain learner fed the same stream (clip norms are summed in different orders)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
forms = [("fp32", dict()), ("fp32", dict(per_layer=True)), ("fp32", dict(shard_opt=True)), ("fp16", dict(half_grads=True)),
         ("fp16", dict(half_grads=True, shard_opt=True)), ("fp16", dict(shard_opt=True))]
for precision, kw in forms:
    B = 256
    d = pkg.DQN(58, minibatch=B, hidden=(1024,) * 4, memory=60000, seed=3, use_graph=True, precision=precision, dp_world=1, dp_rank=0)
    d.dp_init(pkg.DQN.dp_unique_id(), **kw)
    env = pkg.EnvFrontEnd(d, 64, max_steps=500, p_end=0.01, seed=9)
    t0 = time.time(); n_upd = 0
    for it in range(150):
        env.step(max(0.1, 1.0 - it / 100.0), 50)
        if d.memory_size() >= 1000:
            for _ in range(20):
                d.dp_update(None); n_upd += 1
            loss, q = d.read_stats()
            assert np.isfinite(loss) and np.isfinite(q), (it, loss, q)
    assert d.dp_graph_active()
    if kw.get("shard_opt"):
        d.dp_gather_state()
    for net in range(4):
        assert np.isfinite(d.get_params(net)).all()
    for kind in (1, 2):
        for net in (0, 1):
            assert np.isfinite(d.get_params(net, kind)).all()
    print(precision, kw, "updates", n_upd, "memory", d.memory_size(), "loss %.4g avg_q %.4g" % (loss, q), "skipped", d.skipped_steps(),
          "%.1fs" % (time.time() - t0), flush=True)
    assert d.actor_iter() == n_upd and d.skipped_steps() == 0
    env.close(); d.close()
print("dp soak OK")
