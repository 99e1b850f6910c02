"""Shared construction of a (HIP learner, C oracle) pair on identical weights/replay."""
import numpy as np

from oracle import c_oracle, torch_ref
from synth import synth_replay


def make_pair(pkg, B=32, S=59, hidden=(1024, 512, 256, 128), n_replay=2048, seed=1, wscale=10.0,
              capacity=None, mean_len=20, **kw):
    rng = np.random.default_rng(seed)
    capacity = capacity or max(4096, n_replay + 1)
    dqn = pkg.DQN(S, minibatch=B, hidden=hidden, memory=capacity, seed=seed, **kw)
    okw = {k: v for k, v in kw.items() if k in ("gamma", "beta", "tau", "soft_update_freq")}
    if "clip_grad" in kw:
        okw["clip"] = kw["clip_grad"]
    if "actor_lr" in kw:
        okw["lr_actor"] = kw["actor_lr"]
    if "critic_lr" in kw:
        okw["lr_critic"] = kw["critic_lr"]
    orc = c_oracle.Oracle(B=B, S=S, hidden=hidden, capacity=capacity, **okw)
    for net, actor in ((0, True), (1, False)):
        w = torch_ref.init_params_np(rng, S, hidden, actor) * wscale
        dqn.set_params(net, w); dqn.CloneNet(net)
        orc.set_params(net, w); orc.clone_to_target(net)
    data = synth_replay(rng, n_replay, S, mean_len=mean_len)
    dqn.add_transitions_arrays(*data)
    orc.add_transitions(*data)
    return dqn, orc, data, rng
