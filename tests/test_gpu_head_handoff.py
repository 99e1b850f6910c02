"""k_head_bwd's cross-block slab hand-off (small_kernels.hip.h) under UNEVEN load, every word checked.

For shapes whose head gradients have no carrier launch (the fp16 learner below 1024 rows; the fp32 learner's actor heads from 1 383 rows when the
bandwidth-tiled kernel does not apply, e.g. 1 440 rows), the head layer's dW / db are reduced over row chunks by the LAST block to arrive for a column block:
write-through (sc1) slab stores, `s_waitcnt vmcnt(0)`, a barrier, a relaxed agent-scope ticket, sc1 loads — the
"handoff-flag" form MI355X_MICROARCH.md lists as valid on gfx950, with no release / acquire fence (that rests on
ISA semantics, not on the HIP memory model).  The guide's own prescription for such a hand-off: test it under uneven load,
consumer L1-warm, checking every word.  Here: the reduced head gradients of 150 updates, recomputed on the host from the
very operands the kernel read (the stored tower top and the head diffs), while a second learner on another stream keeps
the chip unevenly busy.  A stale or missing slab would show up as a whole row chunk's contribution (~1/8 of an element)."""
import numpy as np
import pytest

from helpers import make_pair
from synth import synth_replay

pytestmark = pytest.mark.gpu


def _head_grads(g, H, heads):
    """(W [n][H], b [n]) blocks of the head layers at the end of a dense Caffe-order gradient vector"""
    out, end = [], g.size
    for n in reversed(heads):
        b = g[end - n:end]; W = g[end - n - n * H:end - n].reshape(n, H); end -= n + n * H
        out.append((W, b))
    return out[::-1]


@pytest.mark.parametrize("precision,B,hidden", [("fp16", 512, (1024, 1024)), ("fp16", 256, (256, 512)), ("fp32", 1440, (128, 64))])
def test_head_gradient_handoff_under_uneven_load(pkg, gpu, precision, B, hidden):
    S = 58
    dqn, orc, data, rng = make_pair(pkg, B=B, S=S, hidden=hidden, n_replay=4096, wscale=3.0, precision=precision)
    noise = pkg.DQN(S, minibatch=256, hidden=(1024, 1024, 1024, 1024), memory=8192, seed=9, use_graph=True)
    noise.add_transitions_arrays(*synth_replay(np.random.default_rng(2), 4096, S))
    H, L = hidden[-1], len(hidden)
    worst = 0.0
    for it in range(150):
        for _ in range(1 + it % 4):                       # uneven: 1..4 full updates of another learner in flight beside ours
            noise.update_async(None)
        idx = rng.integers(0, 4096, size=B)
        dqn.update_phase(0, idx)
        gc = dqn.get_params(1, pkg.KIND_G)
        x3 = dqn.debug_read("act3_%d" % L).astype(np.float64)                      # critic(s, a) tower top, as stored
        dq = (dqn.debug_read("q_train").astype(np.float64) - dqn.debug_read("y").astype(np.float64)) / B
        (Wq, bq), = _head_grads(gc, H, (1,))
        ref = dq @ x3
        worst = max(worst, np.abs(Wq[0] - ref).max() / max(np.abs(ref).max(), 1e-30))
        assert abs(bq[0] - dq.sum()) <= 1e-5 * max(abs(dq.sum()), np.abs(dq).sum() * 1e-2)
        dqn.update_phase(1)
        ga = dqn.get_params(0, pkg.KIND_G)
        x1 = dqn.debug_read("act1_%d" % L).astype(np.float64)                      # actor(s) tower top
        dy = dqn.debug_read("dq_da").astype(np.float64)                            # post-invert head diffs [B][10]
        (Wa, ba), (Wp, bp) = _head_grads(ga, H, (4, 6))
        ref = dy.T @ x1
        got = np.concatenate([Wa, Wp])
        worst = max(worst, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
        np.testing.assert_allclose(np.concatenate([ba, bp]), dy.sum(0), rtol=1e-4, atol=1e-6 * np.abs(dy).sum(0).max())
        dqn.update_phase(2)
    assert worst <= 2e-5, worst            # fp32 sums over the rows in another order; a lost row chunk would be ~1e-1
    assert all(np.isfinite(dqn.read_stats()))
    noise.read_stats(); noise.close(); dqn.close(); orc.close()
