"""Generates the committed golden vectors (run in the BUILD container only; the GPU box just
reads the .npz files).

There is nothing of the reference to import or run here: the reference is C++ on top of Caffe,
which is absent and it ships no fixtures of its own, so parity is
UNPINNED against the reference itself.  What these vectors pin instead: the float64 autograd
restatement (oracle/torch_ref.py, written from the spec, independent of the hand-derived
backward in oracle/dqn_oracle.c).  The explicit inputs SURVEY.md F5 asks for (initial weights, a small replay, the sampled index
lists) are regenerated from a stored seed by the version-independent generators in
tests/synth.py (det_*), with checksums stored to catch drift; each file holds, after each of N updates,
critic_loss, avg_q, q/y vectors, the post-inversion dQ/da, per-blob L2 norms and 16 sampled
elements of every parameter / Adam-state vector, plus GetAction indices for probe states.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import torch_ref  # noqa: E402
from synth import det_params, det_replay, det_indices, det_uniform  # noqa: E402

CASES = {
    # name: (B, S, hidden, wscale, n_replay, n_updates)
    "tiny_B32_S59_h64x64": (32, 59, (64, 64), 8.0, 256, 5),
    "ref_B32_S59_1024_512_256_128": (32, 59, (1024, 512, 256, 128), 5.0, 512, 3),
    "ref_B32_S68_128x4": (32, 68, (128, 64, 64, 64), 5.0, 256, 5),
}


def blob_slices(shapes):
    off = 0
    out = []
    for (n, k) in shapes:
        out.append((off, off + n * k)); off += n * k
        out.append((off, off + n)); off += n
    return out


def main():
    for name, (B, S, hidden, ws, n_rep, n_upd) in CASES.items():
        seed = sum(map(ord, name)) % 1000 + 1
        t = torch_ref.TorchRef(B=B, S=S, hidden=hidden)
        w0 = {}
        for net, actor in ((0, True), (1, False)):
            w = det_params(seed + net, S, hidden, actor, ws)
            w0[net] = w
            t.set_params(net, w); t.set_params(net + 2, w)
        s, a, r, mc, nx, term = det_replay(seed, n_rep, S)
        idxs = det_indices(seed, n_upd, B, n_rep)
        # inputs are NOT stored: they are regenerated from `seed` by tests/synth.py det_*()
        out = dict(B=B, S=S, hidden=np.array(hidden), wscale=ws, seed=seed, n_replay=n_rep, n_updates=n_upd,
                   check_w_actor_sum=np.float64(w0[0].astype(np.float64).sum()),
                   check_replay_sum=np.float64(s.astype(np.float64).sum() + a.astype(np.float64).sum()))
        prng = np.random.default_rng(seed)
        pick = {net: np.sort(prng.choice(w0[net % 2].size, 16, replace=False)) for net in range(4)}
        for u in range(n_upd):
            idx = idxs[u]
            loss, avgq = t.update(s[idx], a[idx], r[idx], mc[idx], nx[idx], term[idx])
            out["u%d_loss" % u] = np.float64(loss); out["u%d_avgq" % u] = np.float64(avgq)
            for k in ("q_target", "y", "q_train", "q_policy", "actor_out", "dq_da"):
                out["u%d_%s" % (u, k)] = t.dbg[k].numpy().astype(np.float64)
            for net in range(4):
                v = t.get_params(net)
                shapes = t.sa if net % 2 == 0 else t.sc
                out["u%d_w%d_norms" % (u, net)] = np.array([np.linalg.norm(v[a0:b0]) for a0, b0 in blob_slices(shapes)])
                out["u%d_w%d_pick" % (u, net)] = v[pick[net]]
            for net in (0, 1):
                out["u%d_m%d_pick" % (u, net)] = t.get_params(net, 1)[pick[net]]
                out["u%d_v%d_pick" % (u, net)] = t.get_params(net, 2)[pick[net]]
        for net in range(4):
            out["pick%d" % net] = pick[net]
        # GetAction probe on the FINAL actor
        probe = det_uniform(seed * 10 + 9, 128 * S, -1, 1).reshape(128, S).astype(np.float32)
        with torch.no_grad():
            ao = t.actor(t.w[0], torch.as_tensor(probe, dtype=torch.float64)).numpy()
        logits = ao[:, :4].copy(); logits[:, 2] = -99999
        out["probe_actor_out"] = ao
        out["probe_action"] = np.argmax(logits, axis=1).astype(np.int32)
        srt = np.sort(ao[:, [0, 1, 3]], axis=1)
        out["probe_margin"] = srt[:, -1] - srt[:, -2]
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB", "min argmax margin %.3g" % out["probe_margin"].min())


if __name__ == "__main__":
    main()
