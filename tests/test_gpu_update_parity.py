"""HIP path vs CPU oracle on identical weights, transitions and sampled indices
(SURVEY.md F5).  Tolerances: Q-values 1e-4 abs (BASELINE.json north_star), action
indices exact."""
import numpy as np
import pytest

from helpers import make_pair
from oracle import c_oracle

pytestmark = pytest.mark.gpu

QTOL = 1e-4   # north_star: "Q-values within 1e-4 fp32" (absolute, for |Q| = O(1))
QRTOL = 1e-5  # plus 1e-5 relative so the bound stays meaningful when the random nets emit |Q| >> 1


def _fro(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


PER_FLIP = 5e-4   # what one unit on the other side of zero may add to a gradient's relative Frobenius error (see _sign_flips, _check_grads)


def _sign_flips(dqn, orc, p, which, same_weights=False):
    """(mismatches, rows) between the SIGNS of the stored tower activations of the learner's pass p and the oracle's last
    actor ('A') / critic ('C') forward.  ReLU is applied in place (src/dqn.cpp:409-410), so ReLU' is taken from the sign of
    the stored output; a pre-activation within fp32 round-off of zero can land on either side — on this library about once
    in ten updates at 256 x 4 x 1024 units per pass, almost surely at 4096 rows — and then that unit's derivative for that row
    is 1 on one side and 0.01 on the other: ~1e-4..1e-3 of the gradient norm in the layers below it.  Counting them makes
    the exception explicit instead of resting on flip-free seeds."""
    n, rows = 0, set()
    for i in range(1, len(orc.hidden) + 1):
        a, b = dqn.debug_read("act%d_%d" % (p, i)), orc.debug_read("act%s_%d" % (which, i))
        bad = (a > 0) != (b > 0)
        if bad.any() and same_weights:
            # both sides evaluated this pass with IDENTICAL weights: a legitimate flip is then a pre-activation within fp32 round-off
            # of zero on BOTH sides (the stored value is x or 0.01 x) — anything larger with the wrong sign is a bug, not round-off.
            # (Not applicable to a pass whose weights already differ by an Adam step, e.g. the policy pass after a flip in Step(1).)
            scale = float(np.abs(b).max())
            assert float(np.abs(a[bad]).max()) <= 2e-5 * scale and float(np.abs(b[bad]).max()) <= 2e-5 * scale, \
                (p, i, float(np.abs(a[bad]).max()), float(np.abs(b[bad]).max()), scale)
        n += int(bad.sum()); rows.update(np.nonzero(bad.any(axis=1))[0].tolist())
    return n, rows


def _check_grads(dqn, orc, net, ref64=None, flips=0):
    """Raw gradients (before clip/Adam), Frobenius-relative: 1e-5 against the C oracle AND against the float64 autograd
    restatement at every shape including BASELINE's, whenever the two sides differentiated the same ReLU branches (flips == 0:
    most updates at every shape); 1e-5 + PER_FLIP per flipped unit otherwise.

    (Rounds 1-3 needed 5e-3 against the C oracle at 4 x 1024 regardless: its GEMMs were k-ordered fp32 fmaf chains whose
    own round-off flipped about one unit per update relative to float64.  The oracle now accumulates every dot product in
    double and rounds once — the order-free value of an sgemm — and the two comparators agree.)"""
    # flip-proportional, no cap (round 6): a flipped unit moves the gradient by up to ~1e-4 of its norm (measured 1.4e-5 … 1.1e-4 at
    # 256 x 4 x 1024, tests/test_gpu_multiseed_parity.py); one in Step(1) moves a few critic weights by another Adam step, after which
    # the policy pass differs in tens of near-zero units — all counted in `flips`
    tol = 1e-5 + flips * PER_FLIP
    g1, g2 = dqn.get_params(net, 3), orc.grad_view(net).copy()
    if ref64 is not None:
        assert _fro(g1, ref64) <= tol, (net, _fro(g1, ref64), flips)
    assert _fro(g1, g2) <= tol, (net, _fro(g1, g2), flips)


def _check_update(dqn, orc, idx, t64=None, data=None):
    """One update, phase by phase (the phases are where gradients are complete)."""
    g64 = [None, None]
    if t64 is not None:
        s, a, r, mc, nx, term = data
        t64.update(s[idx], a[idx], r[idx], mc[idx], nx[idx], term[idx])
        g64 = [t64.g[0].numpy(), t64.g[1].numpy()]
    dqn.update_phase(0, idx); orc.update_phase(0, idx)
    f_c1, _ = _sign_flips(dqn, orc, 3, "C", same_weights=True)         # critic(s, a): the training forward
    _check_grads(dqn, orc, 1, g64[1], f_c1)        # critic dW/db of Step(1)
    dqn.update_phase(1); orc.update_phase(1, idx)
    f_a, rows_a = _sign_flips(dqn, orc, 1, "A", same_weights=True)     # actor(s)
    f_c2, rows_c2 = _sign_flips(dqn, orc, 4, "C", same_weights=f_c1 == 0)   # critic(s, mu(s)) with the updated critic
    _check_grads(dqn, orc, 0, g64[0], f_c1 + f_a + f_c2)        # actor dW/db (a flip in the critic's step moves its weights, hence everything after)
    dqn.update_phase(2); orc.update_phase(2, idx)
    l1, q1 = dqn.read_stats()
    l2, q2 = orc.last_stats()
    for name in ("q_target", "y", "q_train", "q_policy"):
        np.testing.assert_allclose(dqn.debug_read(name), orc.debug_read(name), rtol=QRTOL, atol=QTOL, err_msg=name)
    np.testing.assert_array_equal(dqn.debug_read("idx").astype(np.int64), np.asarray(idx, np.int64))
    np.testing.assert_array_equal(dqn.debug_read("terminal"), orc.debug_read("terminal"))
    a1, a2 = dqn.debug_read("actor_out"), orc.debug_read("actor_out")
    np.testing.assert_allclose(a1, a2, rtol=1e-4, atol=1e-4)
    # post-invert dQ/da: 1e-4 of its scale element by element, in every row whose critic(s, mu(s)) pass took the same branches
    ok = np.ones(len(a1), bool); ok[list(rows_c2)] = False
    if f_c1:
        ok[:] = False                                # the critic's own step differed: every row's dQ/da moves a little
    g1, g2 = dqn.debug_read("dq_da"), orc.debug_read("dq_da")
    assert np.abs(g1 - g2)[ok].max(initial=0.0) <= 1e-4 * max(np.abs(g2).max(), 1e-30), ("dq_da", np.abs(g1 - g2)[ok].max(), np.abs(g2).max())
    assert np.abs(g1 - g2).max() <= 0.2 * np.abs(g2).max()
    assert abs(l1 - l2) <= 1e-4 * max(1.0, abs(l2)), (l1, l2)
    assert abs(q1 - q2) <= QTOL + QRTOL * abs(q2), (q1, q2)
    if t64 is not None:
        for name in ("q_target", "y", "q_train", "q_policy"):
            np.testing.assert_allclose(dqn.debug_read(name), t64.dbg[name].numpy(), rtol=QRTOL, atol=QTOL, err_msg=name)
        # against float64 the per-row quantities are TIGHT: mu(s) to 1e-5 of its scale, the post-invert dQ/da to 1e-4 of its
        # scale per element (same rows as above: float64 sits where the oracle sits)
        ref = t64.dbg["actor_out"].numpy()
        assert np.abs(a1 - ref).max() <= 1e-5 * np.abs(ref).max(), ("actor_out", np.abs(a1 - ref).max())
        ref = t64.dbg["dq_da"].numpy()
        assert np.abs(g1 - ref)[ok].max(initial=0.0) <= 1e-4 * max(np.abs(ref).max(), 1e-30), ("dq_da vs float64", np.abs(g1 - ref)[ok].max())
    return f_c1 + f_a + f_c2


@pytest.mark.parametrize("shape", [
    dict(B=32, S=59, hidden=(1024, 512, 256, 128), f64=True),   # reference defaults (src/dqn.hpp:19, dqn.cpp:425)
    dict(B=32, S=68, hidden=(128, 64, 64, 64), f64=True),  # 1v1 state size, small tower
    dict(B=64, S=77, hidden=(256, 128), f64=True),         # 2v1 state size, 2-layer tower
    dict(B=1024, S=58, hidden=(256, 256), wscale=3.0),     # large minibatch: the bandwidth-tiled head kernels
    dict(B=96, S=61, hidden=(192, 320, 64), wscale=5.0, f64=True),   # ragged: no dimension a multiple of 128, 3 layers
    dict(B=32, S=120, hidden=(64,), wscale=5.0),           # one hidden layer, 130-wide critic input (3 K panels)
    dict(B=64, S=59, hidden=(128, 64, 64, 128, 64, 64), wscale=8.0),   # six layers
    dict(B=32, S=58, hidden=(2048, 1536), wscale=2.0),     # wider than the 1024-column head strips
    dict(B=4096, S=58, hidden=(256, 256), wscale=3.0),     # config #5's minibatch on the fp32 path
    # BASELINE.json config #2.  wscale 2 (weights N(0, 0.02^2)): at 5x the 4x1024 critic's loss
    # explodes to 5e4 after one lr=1e-3 Adam step and HIP, the C oracle and a float64 reference
    # then differ from each other by ReLU-mask flips in different rows (all three measured).
    # No seed is picked here any more (rounds 4-5 scanned for a flip-free one): an update at this size has a ~25 % chance of a
    # round-off flip whatever the summation order; flips are counted, checked to be round-off (|x| <= 2e-5 of the layer's scale on
    # both sides, _sign_flips) and bounded per update; the statement over MANY seeds is tests/test_gpu_multiseed_parity.py
    dict(B=256, S=58, hidden=(1024, 1024, 1024, 1024), wscale=2.0, f64=True),
    # minibatches above 256 rows on a second seed each
    dict(B=1024, S=58, hidden=(256, 256), wscale=3.0, seed=2),
    dict(B=512, S=58, hidden=(1024, 1024), wscale=2.0, seed=2, f64=True),
])
def test_update_matches_oracle(pkg, gpu, shape):
    shape = dict(shape)
    shape.setdefault("wscale", 5.0)
    use64 = shape.pop("f64", False)
    no_flips = shape.pop("no_flips", False)
    dqn, orc, data, rng = make_pair(pkg, n_replay=2048, **shape)
    B = shape["B"]
    t64 = None
    if use64:
        from oracle import torch_ref
        t64 = torch_ref.TorchRef(B=B, S=shape["S"], hidden=shape["hidden"])
        for net in range(4):
            t64.set_params(net, orc.get_params(net))
    n_it = 3 if use64 else 4
    flips = 0
    for it in range(n_it):
        idx = rng.integers(0, 2048, size=B)
        f = _check_update(dqn, orc, idx, t64, data)
        flips += f
        if f:       # the two sides took different branches: their states now differ by a legitimate O(lr) in a few elements,
            #         which would put MORE near-zero units on different sides next time — continue from identical states
            for ref in [orc] + ([t64] if t64 is not None else []):
                for net in range(4):
                    ref.set_params(net, dqn.get_params(net))
                for kind in (1, 2):
                    for net in (0, 1):
                        ref.set_params(net, dqn.get_params(net, kind), kind)
    print("flipped units over %d updates: %d" % (n_it, flips))
    if B <= 128 or no_flips:
        assert flips == 0, flips          # (small shapes: so the 1e-5 bound was the one applied, e.g. at the reference's shape)
    # Adam's normalised step m/(sqrt(v)+eps) is O(1) whatever |g| is, so an element whose gradient
    # is at fp32-roundoff level may legitimately move differently by up to lr per update; on
    # average the parameters must agree to 1% of a step.
    lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
    for net in range(4):
        refs = ([t64.get_params(net)] if use64 else []) + [orc.get_params(net)]
        for ref in refs:
            d = np.abs(dqn.get_params(net) - ref)
            assert d.max() <= n_it * lr[net] + 1e-6, (net, d.max())
            assert d.mean() <= 0.01 * lr[net] + 1e-8, (net, d.mean())
    for kind in (1, 2):   # Adam m, v
        for net in (0, 1):
            a, b = dqn.get_params(net, kind), (t64 if use64 else orc).get_params(net, kind)
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * np.abs(b).max())
    assert dqn.actor_iter() == n_it and dqn.critic_iter() == n_it
    dqn.close(); orc.close()


def test_action_indices_bit_exact(pkg, gpu):
    """GetAction argmax over {DASH,TURN,KICK} on 512 probe states: indices identical."""
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(1024, 512, 256, 128), wscale=10.0)
    probe = data[0][:512]
    out_h = dqn.SelectActionGreedily(probe)
    out_o = orc.actor_forward(probe)
    assert np.abs(out_h - out_o).max() <= QTOL
    act_o, a1_o, a2_o = c_oracle.get_action(out_o)
    acts_h = [pkg.GetAction(o) for o in out_h]
    assert [a.action for a in acts_h] == list(act_o)
    assert 2 not in act_o                      # TACKLE is never returned (src/dqn.cpp:198)
    # margin between best and second-best logit, so the equality above is not luck
    lg = out_o[:, [0, 1, 3]]
    srt = np.sort(lg, axis=1)
    assert (srt[:, -1] - srt[:, -2]).min() > 10 * np.abs(out_h - out_o).max()
    dqn.close(); orc.close()


def test_deterministic_replay(pkg, gpu):
    """Same weights + same indices twice -> bit-identical parameters (no atomics, fixed
    reduction trees)."""
    res = []
    for rep in range(2):
        dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64), seed=5)
        for it in range(3):
            dqn.UpdateActorCritic(rng.integers(0, 2048, size=32))
        res.append([dqn.get_params(n).copy() for n in range(4)])
        dqn.close(); orc.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)


def test_critic_forward_and_evaluate(pkg, gpu):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(128, 64, 64, 64))
    s, a = data[0][:100], data[1][:100]
    np.testing.assert_allclose(dqn.CriticForward(s, a), orc.critic_forward(s, a), atol=QTOL)
    assert abs(dqn.EvaluateAction(s[0], a[0]) - orc.critic_forward(s[:1], a[:1])[0]) <= QTOL
    dqn.close(); orc.close()
