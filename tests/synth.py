"""Synthetic HFO-like inputs shared by tests and bench."""
import numpy as np


def synth_states(rng, n, S):
    """i.i.d. U(-1,1); idx 12 & 54 in {-1,+1}; (13,14),(51,52) = (sin,cos) of U(-pi,pi)."""
    s = rng.uniform(-1, 1, size=(n, S)).astype(np.float32)
    if S >= 56:
        s[:, 12] = rng.choice([-1.0, 1.0], size=n)
        s[:, 54] = rng.choice([-1.0, 1.0], size=n)
        for (i, j) in ((13, 14), (51, 52)):
            th = rng.uniform(-np.pi, np.pi, size=n)
            s[:, i] = np.sin(th); s[:, j] = np.cos(th)
    return s


def synth_actions(rng, n):
    """GetRandomActorOutput distribution (src/dqn.cpp:664-682)."""
    a = np.empty((n, 10), np.float32)
    a[:, 0:4] = rng.uniform(-1, 1, size=(n, 4))
    a[:, 4] = rng.uniform(-100, 100, size=n)
    a[:, 5:8] = rng.uniform(-180, 180, size=(n, 3))
    a[:, 8] = rng.uniform(0, 100, size=n)
    a[:, 9] = rng.uniform(-180, 180, size=n)
    return a


def synth_replay(rng, n, S, gamma=0.99, mean_len=100, cap_len=500):
    """n transitions in episodes of geometric length; rewards U(-.1,.1), +5 on half of
    the terminal steps; mc targets via the LabelTransitions recurrence."""
    s = synth_states(rng, n + 1, S)
    a = synth_actions(rng, n)
    r = rng.uniform(-0.1, 0.1, size=n).astype(np.float32)
    term = np.zeros(n, np.uint8)
    i = 0
    while i < n:
        ln = int(min(cap_len, max(1, rng.geometric(1.0 / mean_len))))
        e = min(n, i + ln) - 1
        term[e] = 1
        if rng.uniform() < 0.5:
            r[e] += 5.0
        i = e + 1
    nx = s[1:].copy()
    nx[term.astype(bool)] = 0
    mc = np.empty(n, np.float32)
    for i in range(n - 1, -1, -1):
        if term[i] or i == n - 1:
            mc[i] = r[i]
        else:
            mc[i] = np.float32(np.float64(r[i]) + gamma * np.float64(mc[i + 1]))
    return s[:n].copy(), a, r, mc, nx, term


# ---- version-independent deterministic generators (golden fixtures store only seeds) -------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def det_uniform(seed, n, lo=0.0, hi=1.0):
    """n doubles in [lo, hi): splitmix64 of (seed, index); identical on every numpy version."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        bits = _splitmix64(idx)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def det_normal(seed, n):
    """Box-Muller on det_uniform."""
    u1 = det_uniform(seed * 2 + 1, n, 1e-12, 1.0)
    u2 = det_uniform(seed * 2 + 2, n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def det_params(seed, S, hidden, actor, wscale=1.0):
    """gaussian(0.01 * wscale) weights, zero biases, dense Caffe order."""
    in_dim = S if actor else S + 10
    heads = (4, 6) if actor else (1,)
    parts, k, sub = [], in_dim, 0
    for n in tuple(hidden) + heads:
        kk = k if n in hidden or True else k
        parts.append((det_normal(seed * 100 + sub, n * k) * 0.01 * wscale).astype(np.float32)); sub += 1
        parts.append(np.zeros(n, np.float32))
        if sub <= len(hidden):
            k = n
    return np.concatenate(parts)


def det_replay(seed, n, S, gamma=0.99, term_every=9):
    """Deterministic replay: states/actions from det_uniform, every `term_every`-th step
    terminal (+5 reward on every other terminal), mc via the LabelTransitions recurrence."""
    s = det_uniform(seed * 10 + 1, (n + 1) * S, -1, 1).reshape(n + 1, S).astype(np.float32)
    a = np.empty((n, 10), np.float32)
    u = det_uniform(seed * 10 + 2, n * 10).reshape(n, 10)
    a[:, 0:4] = u[:, 0:4] * 2 - 1
    a[:, 4] = u[:, 4] * 200 - 100
    a[:, 5:8] = u[:, 5:8] * 360 - 180
    a[:, 8] = u[:, 8] * 100
    a[:, 9] = u[:, 9] * 360 - 180
    r = det_uniform(seed * 10 + 3, n, -0.1, 0.1).astype(np.float32)
    term = np.zeros(n, np.uint8)
    term[term_every - 1::term_every] = 1
    term[-1] = 1
    r[np.where(term)[0][::2]] += 5.0
    nx = s[1:].copy(); nx[term.astype(bool)] = 0
    mc = np.empty(n, np.float32)
    for i in range(n - 1, -1, -1):
        mc[i] = r[i] if (term[i] or i == n - 1) else np.float32(np.float64(r[i]) + gamma * np.float64(mc[i + 1]))
    return s[:n].copy(), a, r, mc, nx, term


def det_indices(seed, n_updates, B, n_replay):
    u = det_uniform(seed * 10 + 7, n_updates * B).reshape(n_updates, B)
    return np.minimum((u * n_replay).astype(np.int64), n_replay - 1)
