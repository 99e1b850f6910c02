#!/usr/bin/env python
"""One rank of a NATIVE data-parallel group (RCCL inside libdqnhip.so, include/dqnhip.h dqnhip_dp_*): the process the
multi-GPU tests and bench.py's graph probe start once per GPU.  No torch, no launcher: the ranks find each other through
dqnhip_dp_init_file (a file rendezvous), so it runs wherever `python` and the built library are.

  python tests/dp_native_worker.py --rank R --world W --device D --rv /tmp/path --out result.json
         [--precision fp32|fp16] [--per-layer] [--half] [--shard-opt] [--rows 64] [--hidden 256,128,64,64]
         [--updates 3] [--mode parity|probe]

mode parity (tests/test_gpu_dp_native.py):
  (a) an EAGER group member runs `updates` updates on explicit per-rank indices; rank 0 also runs ONE plain learner on the
      concatenated minibatch (all shards, same indices shifted per shard) and reports how far the group is from it;
  (b) a graph-replaying member and a second eager member (device-side sampling: same seed, same counters -> same rows)
      run `updates + 2` updates and must hold the same bits after every one (collectives inside the captured graph);
  every rank reports a digest of its replicas — the parent compares them across ranks.
mode probe (bench.py --gpus N): only (b), at the bench's shape — does the captured data-parallel update complete with THIS
  many ranks?  The parent kills the process if it does not; nothing hangs the benchmark itself.

What it checks is the reference's own update (src/dqn.cpp:828-972) split over ranks as SURVEY 8(e) prescribes: global-B
EuclideanLoss normaliser, un-normalised actor gradient sum (:918-921), clip on the reduced gradient, identical Adam step on
every rank.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

N_SHARD = 1024


def digest(d):
    h = hashlib.sha256()
    for net in range(4):
        h.update(np.ascontiguousarray(d.get_params(net)).tobytes())
    for kind in (1, 2):
        for net in (0, 1):
            h.update(np.ascontiguousarray(d.get_params(net, kind)).tobytes())
    return h.hexdigest()


def init_params(rng, in_dim, hidden, heads, wscale):
    """gaussian(std 0.01 * wscale) weights, zero biases (src/dqn.cpp:350-352), dense Caffe order: (W[n][k], b[n]) per layer;
    every head reads the tower top"""
    parts, k = [], in_dim
    for n in tuple(hidden) + tuple(heads):
        parts.append((rng.standard_normal((n, k)) * 0.01 * wscale).astype(np.float32).ravel())
        parts.append(np.zeros(n, np.float32))
        if len(parts) // 2 <= len(hidden):
            k = n
    return np.concatenate(parts)


def fro(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--rv", required=True, help="rendezvous path prefix (shared by the ranks of this run)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"])
    ap.add_argument("--per-layer", action="store_true")
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--shard-opt", action="store_true")
    ap.add_argument("--rows", type=int, default=64, help="minibatch rows per rank")
    ap.add_argument("--hidden", default="256,128,64,64")
    ap.add_argument("--state-size", type=int, default=59)
    ap.add_argument("--updates", type=int, default=3)
    ap.add_argument("--wscale", type=float, default=5.0)
    ap.add_argument("--mode", default="parity", choices=["parity", "probe"])
    ap.add_argument("--timeout", type=int, default=60, help="rendezvous timeout, seconds")
    args = ap.parse_args()

    from __graft_entry__ import load_package
    from synth import synth_replay
    pkg = load_package()
    rank, world = args.rank, args.world
    dev = args.device if args.device is not None else rank
    hid = tuple(int(x) for x in args.hidden.split(","))
    S, Bl = args.state_size, args.rows
    res = {"rank": rank, "world": world, "device": dev, "mode": args.mode, "precision": args.precision,
           "per_layer": args.per_layer, "half": args.half, "shard_opt": args.shard_opt, "ok": False}

    def write():
        tmp = args.out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(res, f)
        os.replace(tmp, args.out)

    write()
    rng = np.random.default_rng(3)
    w = [init_params(rng, S if a else S + 10, hid, (4, 6) if a else (1,), args.wscale) for a in (True, False)]
    shards = [synth_replay(np.random.default_rng(10 + r), N_SHARD, S, mean_len=10) for r in range(world)]
    dp_kw = dict(per_layer=args.per_layer, half_grads=args.half)
    if args.shard_opt:
        dp_kw["shard_opt"] = True
    if world > 1 and (args.per_layer or args.shard_opt):
        dp_kw["unverified_ok"] = True          # these forms have never met a second rank: this run is what would verify them

    def member(use_graph, tag):
        d = pkg.DQN(S, minibatch=Bl, hidden=hid, memory=4096, seed=7, device=dev, dp_world=world, dp_rank=rank,
                    precision=args.precision, use_graph=use_graph)
        for net in (0, 1):
            d.set_params(net, w[net]); d.CloneNet(net)
        d.add_transitions_arrays(*shards[rank])
        d.dp_init_file("%s.%s" % (args.rv, tag), timeout_s=args.timeout, **dp_kw)
        return d

    t0 = time.time()
    idx_rng = np.random.default_rng(99)
    idx_all = idx_rng.integers(0, N_SHARD, size=(args.updates, world, Bl))
    if args.mode == "parity":
        e = member(False, "eager")
        stats = []
        for u in range(args.updates):
            e.dp_update(idx_all[u, rank])
            stats.append(e.read_stats())
        res["eager_stats"] = stats
        if args.shard_opt:
            e.dp_gather_state()           # (collective) the Adam history of the other ranks' slices
        res["eager_digest"] = digest(e)
        res["iters"] = [e.actor_iter(), e.critic_iter()]
        if rank == 0:
            one = pkg.DQN(S, minibatch=Bl * world, hidden=hid, memory=world * N_SHARD + 1, seed=7, device=dev, precision=args.precision)
            for net in (0, 1):
                one.set_params(net, w[net]); one.CloneNet(net)
            for sh in shards:
                one.add_transitions_arrays(*sh)
            one_stats = []
            for u in range(args.updates):
                gidx = np.concatenate([idx_all[u, r] + r * N_SHARD for r in range(world)])
                one_stats.append(one.UpdateActorCritic(gidx))
            res["one_stats"] = one_stats
            lr = {0: 1e-5, 1: 1e-3, 2: 1e-5 * 1e-3, 3: 1e-3 * 1e-3}
            dev_from_one = {}
            for net in range(4):
                dd = np.abs(e.get_params(net) - one.get_params(net))
                dev_from_one["w%d" % net] = [float(dd.max() / lr[net]), float(dd.mean() / lr[net])]     # in Adam steps
            for kind, nm in ((1, "m"), (2, "v")):
                for net in (0, 1):
                    dev_from_one["%s%d" % (nm, net)] = fro(e.get_params(net, kind), one.get_params(net, kind))
            # the last update's REDUCED gradients are still in the arenas
            for net in (0, 1):
                dev_from_one["g%d" % net] = fro(e.get_params(net, pkg.KIND_G), one.get_params(net, pkg.KIND_G))
            res["vs_single_learner"] = dev_from_one
            one.close()
        e.close()
    # (b) graph replay against eager, device-side sampling
    g, e2 = member(True, "graph"), member(False, "eager2")
    n_b = args.updates + 2
    same = True
    for u in range(n_b):
        # one communicator's collective at a time on this device: two groups' RCCL kernels enqueued concurrently on
        # different streams may start in different orders on different ranks (the classic multi-communicator hazard)
        g.dp_update(None); sg = g.read_stats()
        e2.dp_update(None); se = e2.read_stats()
        same = same and (sg == se)
    res["graph_active"] = bool(g.dp_graph_active())
    if args.shard_opt:
        g.dp_gather_state(); e2.dp_gather_state()
    res["single_graph_equals_eager"] = bool(same and digest(g) == digest(e2))
    write()                                  # (a parent that has to kill this process during the next stage still learns this much)
    # dqnhip_dp_update_n: sixteen updates (collectives included) per hipGraph launch, gathers riding ahead — against single
    # eager updates (one communicator at a time: read_stats blocks)
    n_multi = 16 + 3
    g.dp_update_n(n_multi); sg = g.read_stats()
    for _ in range(n_multi):
        e2.dp_update(None)
    se = e2.read_stats()
    multi_same = (sg == se)
    if args.shard_opt:
        g.dp_gather_state(); e2.dp_gather_state()
    res["graph_equals_eager"] = bool(same and multi_same and digest(g) == digest(e2))
    res["multi_equals_eager"] = bool(multi_same and digest(g) == digest(e2))
    res["graph_digest"] = digest(g)
    res["graph_stats"] = list(g.read_stats())
    # how long one captured update takes with this many ranks (the probe's by-product)
    if args.mode == "probe":
        g.read_stats()
        t1 = time.time()
        for _ in range(48):
            g.dp_update(None)
        g.read_stats()
        res["graph_ms_per_update"] = (time.time() - t1) / 48 * 1e3
        t1 = time.time()
        g.dp_update_n(48)
        g.read_stats()
        res["graph_n_ms_per_update"] = (time.time() - t1) / 48 * 1e3
    g.close(); e2.close()
    res["seconds"] = time.time() - t0
    res["ok"] = True
    write()


if __name__ == "__main__":
    main()
