"""Independent restatement of the reference's update in PyTorch (CPU), written from
SURVEY.md §3.3/§8c rather than from dqn_oracle.c: gradients come from autograd,
not hand-written backward passes.  TEST INFRASTRUCTURE ONLY (tests/, smoke(),
bench.py's cpu_baseline leg).  PARITY UNPINNED (see dqn_oracle.c header).

Two uses:
  * dtype=float64: cross-checks the C restatement (tests/test_oracle_cross.py) and
    generates the committed golden vectors (tests/golden/make_golden.py);
  * dtype=float32: "CPU-B" of BASELINE.md §3 — the same op sequence with MKL/oneDNN
    GEMMs, the closest available proxy for Caffe + an optimised BLAS.

Reference lines followed: src/dqn.cpp:828-972 (sequence), :418-454 (topology),
:893-900 (target), :918-957 (actor gradient, inverting gradients), :1085-1096
(soft update); Caffe semantics S1-S12 of SURVEY.md §8c.
"""
import math

import numpy as np
import torch

NA, NP_, NO = 4, 6, 10
SLOPE = 0.01


def layout(in_dim, hidden, heads):
    """dense Caffe learnable_params order: (W[n_out, k], b[n_out]) per layer."""
    shapes = []
    k = in_dim
    for h in hidden:
        shapes.append((h, k)); k = h
    for h in heads:
        shapes.append((h, k))
    return shapes


def unpack(vec, shapes):
    out, off = [], 0
    for (n, k) in shapes:
        W = vec[off:off + n * k].view(n, k); off += n * k
        b = vec[off:off + n]; off += n
        out.append((W, b))
    assert off == vec.numel()
    return out


def mlp(x, params, n_tower):
    for i in range(n_tower):
        W, b = params[i]
        x = torch.nn.functional.leaky_relu(x @ W.t() + b, SLOPE)
    return torch.cat([x @ W.t() + b for (W, b) in params[n_tower:]], dim=1)


class TorchRef:
    def __init__(self, B=32, S=59, hidden=(1024, 512, 256, 128), gamma=0.99, beta=0.5, tau=0.001,
                 soft_update_freq=1, lr_actor=1e-5, lr_critic=1e-3, beta1=0.95, beta2=0.999,
                 eps=1e-8, clip=10.0, dtype=torch.float64, global_B=None):
        self.B, self.S, self.hidden, self.dtype = B, S, tuple(hidden), dtype
        self.L = len(hidden)
        self.gamma, self.beta = gamma, beta
        self.tau = float(np.float32(tau))           # passed as float (src/dqn.cpp:968)
        self.freq = soft_update_freq
        f32 = lambda v: float(np.float32(v))         # solver fields are proto floats
        self.lr = [f32(lr_actor), f32(lr_critic)]
        self.b1, self.b2, self.eps, self.clip = f32(beta1), f32(beta2), f32(eps), f32(clip)
        self.global_B = global_B or B
        self.sa = layout(S, hidden, (NA, NP_))
        self.sc = layout(S + NO, hidden, (1,))
        na = sum(n * k + n for n, k in self.sa)
        nc = sum(n * k + n for n, k in self.sc)
        z = lambda n: torch.zeros(n, dtype=dtype)
        self.w = [z(na), z(nc), z(na), z(nc)]        # actor, critic, actor_t, critic_t
        self.m = [z(na), z(nc)]
        self.v = [z(na), z(nc)]
        self.g = [z(na), z(nc)]
        self.iter = [0, 0]
        self.dbg = {}

    # parameters ------------------------------------------------------------
    def set_params(self, net, arr, kind=0):
        t = torch.as_tensor(np.asarray(arr), dtype=self.dtype).clone()
        [self.w, self.m, self.v, self.g][kind][net] = t

    def get_params(self, net, kind=0):
        return [self.w, self.m, self.v, self.g][kind][net].detach().numpy().copy()

    def actor(self, vec, s):
        return mlp(s, unpack(vec, self.sa), self.L)

    def critic(self, vec, s, a):
        return mlp(torch.cat([s, a], dim=1), unpack(vec, self.sc), self.L)[:, 0]

    # solver ------------------------------------------------------------------
    def _apply(self, net, grad):
        """ClipGradients + Adam + Net::Update."""
        g = grad.clone()
        if self.clip >= 0:
            l2 = torch.sqrt((g * g).sum())
            if l2 > self.clip:
                g = g * (self.clip / l2)
        t = self.iter[net] + 1
        corr = math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        self.m[net] = self.b1 * self.m[net] + (1 - self.b1) * g
        self.v[net] = self.b2 * self.v[net] + (1 - self.b2) * g * g
        upd = (self.lr[net] * corr) * self.m[net] / (torch.sqrt(self.v[net]) + self.eps)
        self.w[net] = self.w[net] - upd
        self.iter[net] += 1

    # the update ----------------------------------------------------------------
    def update(self, s, a, r, mc, s_next, terminal):
        """One UpdateActorCritic on an explicit minibatch (already gathered)."""
        dt = self.dtype
        s = torch.as_tensor(np.asarray(s), dtype=dt); a = torch.as_tensor(np.asarray(a), dtype=dt)
        r = torch.as_tensor(np.asarray(r), dtype=dt); mc = torch.as_tensor(np.asarray(mc), dtype=dt)
        sn = torch.as_tensor(np.asarray(s_next), dtype=dt)
        term = torch.as_tensor(np.asarray(terminal).astype(bool))
        B = self.B
        with torch.no_grad():
            mu_n = self.actor(self.w[2], sn)
            q_t = self.critic(self.w[3], sn, mu_n)
            off = torch.where(term, r, r + self.gamma * q_t)
            y = self.beta * mc + (1 - self.beta) * off
        # critic step: EuclideanLoss = sum(d^2) / (2 * num)
        wc = self.w[1].clone().requires_grad_(True)
        q = self.critic(wc, s, a)
        loss = ((q - y) ** 2).sum() / (2 * self.global_B)
        (gc,) = torch.autograd.grad(loss, wc)
        self.g[1] = gc.detach().clone()
        self._apply(1, gc.detach())
        # actor step: gradient of -sum(Q) wrt actions, inverted, pushed through the actor
        wa = self.w[0].clone().requires_grad_(True)
        mu = self.actor(wa, s)
        mu_d = mu.detach().clone().requires_grad_(True)
        q2 = self.critic(self.w[1], s, mu_d)
        (dq_da,) = torch.autograd.grad(-q2.sum(), mu_d)
        mn = torch.tensor([-1.] * 4 + [0., -180., -180., -180., 0., -180.], dtype=dt)
        mx = torch.tensor([1.] * 4 + [100., 180., 180., 180., 100., 180.], dtype=dt)
        out = mu.detach()
        inv = torch.where(dq_da < 0, dq_da * (mx - out) / (mx - mn),
                          torch.where(dq_da > 0, dq_da * (out - mn) / (mx - mn), dq_da))
        (ga,) = torch.autograd.grad(mu, wa, grad_outputs=inv)
        self.g[0] = ga.detach().clone()
        self._apply(0, ga.detach())
        if max(self.iter) % self.freq == 0:
            with torch.no_grad():
                self.w[3] = self.tau * self.w[1] + (1 - self.tau) * self.w[3]
                self.w[2] = self.tau * self.w[0] + (1 - self.tau) * self.w[2]
        self.dbg = dict(q_target=q_t, y=y, q_train=q.detach(), q_policy=q2.detach(),
                        actor_out=out, dq_da=inv)
        return float(loss.detach()), float(q2.detach().sum() / self.global_B)


def init_params_np(rng, S, hidden, actor):
    """gaussian(std 0.01) weights, zero bias (src/dqn.cpp:350-352), dense order."""
    shapes = layout(S if actor else S + NO, hidden, (NA, NP_) if actor else (1,))
    parts = []
    for (n, k) in shapes:
        parts.append((rng.standard_normal((n, k)) * 0.01).astype(np.float32).ravel())
        parts.append(np.zeros(n, np.float32))
    return np.concatenate(parts)
