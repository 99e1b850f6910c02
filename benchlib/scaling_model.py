"""The xGMI / RCCL cost model behind `sub_records.*.projection`: no measurement happens here."""
from .flops import tower_weights


# xGMI (MI355X, 8 GPUs fully connected): 7 links per GPU, 153.6 GB/s per link counting both directions = 76.8 GB/s
# each way.  Two bounds for a sum all-reduce of S bytes over N ranks: ONE ring = every byte crosses one
# link per step, 2 (N-1)/N S / b; ALL links busy (N-1 edge-disjoint rings, or direct reduce-scatter + all-gather) =
# that divided by N-1.  Per-collective latency floor on top (launch + 2 (N-1) hops).
XGMI_LINK_GBS_ONE_WAY = 76.8


COLLECTIVE_LATENCY_US = 20.0


def collective_us(nbytes, world, links, factor=2.0):
    """One collective of `nbytes` over `world` ranks: factor 2 = all-reduce (reduce-scatter + all-gather volume),
    1 = a reduce-scatter or an all-gather alone; links = 1 (ONE ring: every byte crosses one link per step) or world - 1
    (all links busy: edge-disjoint rings / direct exchange on the full mesh)."""
    if world < 2:
        return 0.0
    vol = factor * (world - 1) / world * nbytes
    return COLLECTIVE_LATENCY_US + vol / (XGMI_LINK_GBS_ONE_WAY * 1e9 * links) * 1e6


def allreduce_projection_us(nbytes, world):
    if world < 2:
        return {"one_ring_us": 0.0, "all_links_us": 0.0}
    return {"one_ring_us": round(collective_us(nbytes, world, 1), 1), "all_links_us": round(collective_us(nbytes, world, world - 1), 1)}


def exposed_exchange_us(buckets, compute_end_us, world, links):
    """Overlap-aware price of one net's gradient exchange.  buckets = [(ready_us, nbytes), ...] in
    issue order: bucket i may start once its producing launch has ended (ready_us, measured from the start of the net's
    backward chain) AND the previous collective has finished (one communication stream: collectives do not overlap each
    other); the phase's own launches end at compute_end_us.  Returns the microseconds the main stream WAITS at the join."""
    t = 0.0
    for ready, nbytes in buckets:
        t = max(t, ready) + collective_us(nbytes, world, links)
    return max(0.0, t - compute_end_us)


# k_adam_soft on 1/N of a 4x1024 net's arena, Infinity-Cache-resident / evicted (profiles/r04_adam_slice_probe.txt, MI355X):
ADAM_SLICE_US = {1: (19.6, 28.2), 2: (9.2, 18.7), 4: (6.8, 12.0), 8: (5.0, 8.7)}


def dp_projection(world, half, net_bytes, t_rank_ms, t_1gpu_ms, bwd_launch_us, narrow_launch_us, per_layer_slices):
    """What N ranks of this rank shape would take per update, from what ONE GPU can measure + the link model
    (76.8 GB/s one way per link, 7 links per GPU, 20 us per collective).  Four exchange forms per link model:
      single      one all-reduce per net after its backward chain (nothing overlaps: the optimiser needs the clip norm of the
                  WHOLE reduced gradient)
      per_layer   one bucket per tower layer on a communication stream, started when that layer's backward launch has ended;
                  the chain's remaining launches run beside it (exposed_exchange_us) — only the part the chain does not
                  cover is paid, but every bucket pays the per-collective latency
      sharded     ZeRO-1 style: reduce-scatter, a 4-float all-reduce for the clip norm + tails, clip+Adam+soft update on this
                  rank's 1/N slice, all-gather of the updated online AND target weights (the target nets move every update:
                  src/dqn.cpp:967-970 — twice an ordinary model's all-gather volume)
    Returns {form: {"one_ring": ms, "all_links": ms, "speedup_one_ring": x, "speedup_all_links": x}}."""
    out = {}
    adam_full, adam_slice = ADAM_SLICE_US[1][0], ADAM_SLICE_US.get(world, ADAM_SLICE_US[8])[0]
    for form in ("single", "per_layer", "sharded"):
        r = {}
        for name, links in (("one_ring", 1), ("all_links", max(1, world - 1))):
            extra = 0.0
            for nb, slices in zip(net_bytes, per_layer_slices):
                if form == "single" or (form == "per_layer" and half):
                    extra += collective_us(nb, world, links)
                elif form == "per_layer":
                    # backward order: top tower layer first; its bucket is ready when its own launch ends
                    ready, t, bk = [], 0.0, []
                    n_big = len(slices) - 2                    # slices = [layer L-1, ..., layer 1, layer 0, head + tail] in bytes
                    for i, sb in enumerate(slices):
                        t += bwd_launch_us if i < n_big else (narrow_launch_us if i == n_big else 5.0)
                        bk.append((t, sb))
                    extra += exposed_exchange_us(bk, t, world, links)
                else:
                    # all-gather, as built: the updated online weights AND the targets in fp32 (2 x 4 B/param) and, for the fp16
                    # learner, their two fp16 mirrors as well (2 x 2 B/param) — against gradient bytes nb of 4 B/param (fp32
                    # exchange) or 2 B/param (bf16 exchange): 2 x nb resp. 6 x nb.  (A leaner fp16 form — mirrors + the fp32
                    # bias / head slices only — would move 2 x nb; it still loses to the single all-reduce)
                    extra += (collective_us(nb, world, links, 1.0) + collective_us(16, world, links)
                              + collective_us((6 if half else 2) * nb, world, links, 1.0) - (adam_full - adam_slice))
            ms = (t_rank_ms[form] if isinstance(t_rank_ms, dict) else t_rank_ms) + extra * 1e-3
            r[name] = round(ms, 4); r["speedup_" + name] = round(t_1gpu_ms / ms, 2)
        out[form] = r
    return out


def grad_bytes(S_, hidden, half):
    """bytes per update that cross the links: both gradient arenas (critic, then actor) + the fp32 tails"""
    na = sum(tower_weights(S_, hidden)) + sum(hidden) + 10 * hidden[-1] + 10
    nc = sum(tower_weights(S_ + 10, hidden)) + sum(hidden) + hidden[-1] + 1
    per = 2 if half else 4
    return [nc * per + 16, na * per + 16]
