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
