#!/usr/bin/env python
"""bench.py — DQN updates/sec on MI355X (BASELINE.json metric), with the kernel roofline
and the CPU baseline timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one DQN::UpdateActorCritic (reference src/dqn.cpp:828-972): on-device sampling
+ minibatch gather from the device-resident replay, 5 tower forwards, 3 backwards, TD
target, 2x (clip + Adam), 2x soft target update.  Workload = BASELINE.json configs[1]:
1v0 HFO, 4x1024 actor/critic MLP, minibatch 256, replay 1M transitions resident in HBM,
58-dim synthetic states.  N > 1: data-parallel weak scaling — every rank keeps its own
replay shard and a 256-row local minibatch (global minibatch 256*N), gradients summed by
RCCL all-reduce; value is reported in minibatch-256 updates/s (= N x global updates/s).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this platform needs dmabuf IPC (RCCL / cross-process tensors fail with
# "hipIpcGetMemHandle: invalid argument" otherwise); normally already exported — never override it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from __graft_entry__ import load_package  # noqa: E402
from benchlib.flops import (HBM_PEAK_GBS, MFMA_F32_PEAK_TF, MFMA_F16_PEAK_TF, KERNEL_NAMES, TRACE_FAMILY, tower_weights, family_flops,  # noqa: E402,F401
                            family_flops16)
from benchlib.scaling_model import (XGMI_LINK_GBS_ONE_WAY, COLLECTIVE_LATENCY_US, ADAM_SLICE_US, collective_us, allreduce_projection_us,  # noqa: E402,F401
                                    exposed_exchange_us, dp_projection, grad_bytes)
from benchlib.profiling import PMC_SUMMARY, pmc_traffic, live_pmc_traffic, live_kernel_trace  # noqa: E402,F401

HIDDEN = (1024, 1024, 1024, 1024)
B = 256
S = 58
REPLAY = 1_000_000


def probe_captured_dp(rank, world, local_rank, precision, half, per_layer, timeout_s=150):
    """Run tests/dp_native_worker.py --mode probe as a child of this rank (the children of all ranks form their own group).
    Returns {"ok": bool, ...}: ok = the child finished in time, its captured update was active and bit-identical to an
    eager group member's.  The child is killed (exactly that process) on timeout."""
    import subprocess
    import tempfile
    port = os.environ.get("MASTER_PORT", "0")
    rv = os.path.join(tempfile.gettempdir(), "dqnhip_probe_%s" % port)
    out = "%s_rank%d.json" % (rv, rank)
    try:
        os.unlink(out)
    except OSError:
        pass
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_native_worker.py"), "--rank", str(rank), "--world", str(world), "--device", str(local_rank),
           "--rv", rv, "--out", out, "--mode", "probe", "--precision", precision, "--rows", str(B), "--hidden", ",".join(str(h) for h in HIDDEN),
           "--state-size", str(S), "--updates", "3", "--wscale", "1.0", "--timeout", "60"]
    if half:
        cmd.append("--half")
    if per_layer:
        cmd.append("--per-layer")
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        log, _ = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()
        log, _ = p.communicate()
        return _probe_result(out, {"ok": False, "multi_ok": False, "why": "child of rank %d killed after %d s" % (rank, timeout_s), "log": log[-400:]}, t0)
    res = {"ok": False, "multi_ok": False, "why": "child of rank %d exited %d: %s" % (rank, p.returncode, log[-300:])}
    return _probe_result(out, res, t0)


def _probe_result(out, res, t0):
    """ok: the one-update graph was active and bit-identical to eager; multi_ok: so was dqnhip_dp_update_n's sixteen-update graph.
    The child writes its findings stage by stage, so a child killed in the second stage still reports the first."""
    try:
        j = json.load(open(out))
        single = bool(j.get("graph_active") and j.get("single_graph_equals_eager"))
        multi = bool(j.get("ok") and single and j.get("multi_equals_eager"))
        res = {"ok": single, "multi_ok": multi, "seconds": round(time.perf_counter() - t0, 1), "graph_ms_per_update": j.get("graph_ms_per_update"),
               "graph_n_ms_per_update": j.get("graph_n_ms_per_update"),
               "why": "" if multi else "%s; graph_active=%s single_graph_equals_eager=%s multi_equals_eager=%s" % (
                   res.get("why", ""), j.get("graph_active"), j.get("single_graph_equals_eager"), j.get("multi_equals_eager"))}
    except Exception:                 # noqa: BLE001 — reported in "why"
        pass
    return res


def prefill(dqn, n, seed, chunk=131072):
    from synth import synth_replay
    rng = np.random.default_rng(seed)
    done = 0
    while done < n:
        m = min(chunk, n - done)
        dqn.add_transitions_arrays(*synth_replay(rng, m, S))
        done += m


# (the cpu_baseline leg stays in this file: bench.py is the one place outside tests/ and smoke() that may execute oracle/ code)
REF_SHAPE = dict(B=32, S=59, hidden=(1024, 512, 256, 128))      # src/dqn.hpp:19, src/dqn.cpp:425: BASELINE configs[0]


def cpu_baseline(B, S, HIDDEN, budget_s=20.0):
    """CPU stand-ins for 'the reference Caffe CPU solver' (which cannot be built here: Caffe / HFO / boost / glog / gflags /
    protobuf are absent), SURVEY 8(d)'s matrix on this box's host cores, bounded to ~budget_s seconds:
      shapes   BASE (B = 256, S = 58, 4 x 1024: the headline's) and REF (B = 32, S = 59, 1024-512-256-128: the reference's
               compile-time defaults, BASELINE configs[0])
      threads  1 and all usable cores
      ports    (A) oracle/dqn_oracle.c executing the reference's op sequence INCLUDING its wasted work (the critic dW it computes
               and discards, first-layer input gradients) — a checker first: every dot product is accumulated in double in a plain
               loop, so it is NOT a BLAS-class proxy; (B) oracle/torch_ref.py in float32 (MKL / oneDNN GEMMs, autograd: necessary
               work only) — the closest available stand-in for Caffe + an optimised BLAS.
    `value` = the fastest BASE-shape, all-cores leg (what the headline is quoted beside)."""
    from oracle import c_oracle, torch_ref
    from synth import synth_replay
    import torch
    rng = np.random.default_rng(11)
    cores = c_oracle.usable_cores()          # affinity capped by the cgroup quota (16 on the GPU box)
    n_rep = 8192
    legs = [(shape, threads, port) for shape in ("base_b256", "ref_b32") for threads in (cores, 1) for port in ("c_port", "torch_fp32")]
    per_leg = budget_s / len(legs)
    res = {"base_b256": {"c_port": {}, "torch_fp32": {}}, "ref_b32": {"c_port": {}, "torch_fp32": {}}}
    shapes = {"base_b256": dict(B=B, S=S, hidden=HIDDEN), "ref_b32": REF_SHAPE}
    cache = {}
    for shape, threads, port in legs:
        sh = shapes[shape]
        if shape not in cache:
            cache[shape] = (synth_replay(rng, n_rep, sh["S"]), [torch_ref.init_params_np(rng, sh["S"], sh["hidden"], a) for a in (True, False)])
        data, wts = cache[shape]
        if port == "c_port":
            c_oracle.set_threads(threads)
            orc = c_oracle.Oracle(B=sh["B"], S=sh["S"], hidden=sh["hidden"], capacity=n_rep + 1, mirror_waste=1)
            for net in (0, 1):
                orc.set_params(net, wts[net]); orc.clone_to_target(net)
            orc.add_transitions(*data)
            one = lambda: orc.update(rng.integers(0, n_rep, size=sh["B"]))
        else:
            torch.set_num_threads(threads)
            t = torch_ref.TorchRef(B=sh["B"], S=sh["S"], hidden=sh["hidden"], dtype=torch.float32)
            for net in (0, 1):
                t.set_params(net, wts[net]); t.set_params(net + 2, wts[net])
            s_, a_, r_, mc_, nx_, term_ = data

            def one():
                idx = rng.integers(0, n_rep, size=sh["B"])
                t.update(s_[idx], a_[idx], r_[idx], mc_[idx], nx_[idx], term_[idx])
        one()                                   # warm-up
        t0 = time.perf_counter(); n = 0
        while n < 2 or time.perf_counter() - t0 < per_leg:
            one(); n += 1
        res[shape][port]["threads_1" if threads == 1 else "threads_all"] = round(n / (time.perf_counter() - t0), 3)
        if port == "c_port":
            orc.close()
    c_oracle.set_threads(cores); torch.set_num_threads(cores)
    base = {p_: res["base_b256"][p_].get("threads_all", res["base_b256"][p_].get("threads_1")) for p_ in ("c_port", "torch_fp32")}
    best = max(base, key=base.get)
    what = {"c_port": "C restatement (oracle/dqn_oracle.c): the reference's op sequence incl. the critic wgrad it computes and discards, OpenMP; "
                      "double-accumulated dot products (a checker, not a BLAS proxy)",
            "torch_fp32": "PyTorch-CPU fp32 restatement (oracle/torch_ref.py): MKL / oneDNN GEMMs, autograd, necessary work only"}
    return {"value": base[best], "unit": "updates/s", "cores": cores, "threads": cores, "kind": "port",
            "sample": "%s; ~%.0f s per leg of B=256 4x1024 updates on an 8192-transition replay; faster of the two ports at BASE shape, all cores" % (what[best], per_leg),
            "updates_per_s": res,
            "ref_shape": {"B": 32, "S": 59, "hidden": [1024, 512, 256, 128], "what": "BASELINE configs[0] / SURVEY 8(d): the reference's compile-time defaults (src/dqn.hpp:19, src/dqn.cpp:425)",
                          "threads_1": {p_: res["ref_b32"][p_].get("threads_1") for p_ in res["ref_b32"]},
                          "threads_all": {p_: res["ref_b32"][p_].get("threads_all") for p_ in res["ref_b32"]}},
            "note": "stand-in: the reference's Caffe CPU solver cannot be built in this image; cores = CPU affinity capped by the cgroup quota; "
                    "of the two ports only torch_fp32 (optimised GEMMs) is a Caffe+BLAS-class proxy"}


def timed(step, sync, n, warm):
    for _ in range(warm):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    return (time.perf_counter() - t0) / n


def timed_n(d, sync, n, warm):
    """n device-sampled updates of one learner, enqueued the way the headline is (dqnhip_update_async_n)"""
    d.update_async_n(warm)
    sync()
    t0 = time.perf_counter()
    d.update_async_n(n)
    sync()
    return (time.perf_counter() - t0) / n


def env_roofline(S_, workers, us_per_step):
    """a batched env step is one actor forward per worker = 2 * Wa FLOP (tower + heads), fp32 MFMA"""
    wa = sum(tower_weights(S_, HIDDEN)) + 10 * HIDDEN[-1]
    ach = 2.0 * wa * workers / (us_per_step * 1e-6) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
            "flops_per_env_step": 2 * wa}


def sub_records(pkg, par, args, rank, world, local_rank, native, barrier):
    """Per-config records next to the headline (BASELINE.json configs[2] and [4]; strong scaling when N > 1).
    Small replay memories: these time kernels, not prefill."""
    import torch
    out = {}
    GB, n_it = 4096, 100
    if world == 1:
        # configs[4] on ONE GPU: minibatch 4096, fp16 MFMA operands / fp32 accumulate, 4x1024
        for prec in ("fp16", "fp32"):
            d = pkg.DQN(S, minibatch=GB, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, use_graph=True, precision=prec)
            prefill(d, 150000, seed=7)
            dt = timed_n(d, torch.cuda.synchronize, n_it + 4, 24)
            fl = sum((family_flops16 if prec == "fp16" else family_flops)(GB, S, HIDDEN).values())
            peak = MFMA_F16_PEAK_TF if prec == "fp16" else MFMA_F32_PEAK_TF
            out["configs4_1gpu_b4096_%s" % prec] = {"ms_per_update": round(dt * 1e3, 4), "updates_per_s": round(1 / dt, 1),
                                                    "samples_per_s": round(GB / dt), "update_mfma_frac": round(fl / dt / 1e12 / peak, 4)}
            d.read_stats(); d.close()
        # the headline's own shape on the fp16-MFMA learner (configs[4]'s precision at configs[1]'s sizes): not the parity path, for the record
        d = pkg.DQN(S, minibatch=B, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, use_graph=True, precision="fp16")
        prefill(d, 150000, seed=7)
        dt = timed_n(d, torch.cuda.synchronize, 512, 64)
        out["configs1_shape_b256_fp16"] = {"ms_per_update": round(dt * 1e3, 5), "updates_per_s": round(1 / dt, 1),
                                           "note": "fp16 MFMA operands, fp32 accumulate / master weights / Adam / heads; launch-bound (22 GEMM launches of ~6.3 us)"}
        d.read_stats(); d.close()
        # two independent agents (learners) on this one GPU, each on its own stream with its own captured update —
        # how the reference packs a multi-agent team onto one device (one DQN per agent thread, src/dqn_main.cpp:62, 264)
        agents = [pkg.DQN(S, minibatch=B, hidden=HIDDEN, memory=100000, seed=11 + i, device=local_rank, use_graph=True) for i in range(2)]
        for i, d in enumerate(agents):
            prefill(d, 60000, seed=21 + i)

        def both():
            for d in agents:
                d.update_async(None)
        dt = timed(both, torch.cuda.synchronize, 500, 50)

        def both_n(k):
            for lo in range(0, k, 64):            # interleave the agents' enqueues: 64 updates (four graph launches) at a time
                for d in agents:
                    d.update_async_n(min(64, k - lo))
        both_n(64); torch.cuda.synchronize()
        t1 = time.perf_counter(); both_n(512); torch.cuda.synchronize()
        dt_n = (time.perf_counter() - t1) / 512
        out["two_agents_one_gpu_b256_fp32"] = {"aggregate_updates_per_s": round(2 / dt_n, 1), "per_agent_updates_per_s": round(1 / dt_n, 1),
                                               "aggregate_updates_per_s_one_update_per_graph_launch": round(2 / dt, 1),
                                               "note": "independent streams overlap each other's launch floors; dqnhip_update_async_n per agent"}
        for d in agents:
            d.read_stats(); d.close()
        # What launches GROUPED across the two agents (both learners' layer i in one launch) could reach at most:
        # ONE learner with a 512-row minibatch runs exactly those launches — every GEMM launch with twice the rows — and is kinder
        # than the grouped form could be (one weight set instead of two, 64 x 32 tiles over shared weight panels).  Its rate in
        # 256-row updates is the bound; two independent streams (above) already deliver more, so the grouped call is not built.
        d = pkg.DQN(S, minibatch=2 * B, hidden=HIDDEN, memory=100000, seed=11, device=local_rank, use_graph=True)
        prefill(d, 60000, seed=21)
        dt512 = timed_n(d, torch.cuda.synchronize, 512, 64)
        d.read_stats(); d.close()
        out["two_agents_one_gpu_b256_fp32"]["grouped_launch_upper_bound"] = {
            "aggregate_updates_per_s": round(2 / dt512, 1), "ms_per_512_row_update": round(dt512 * 1e3, 5),
            "what": "one learner at minibatch 512 = the launches a two-agent grouped update would make, with ONE weight set: a bound on "
                    "dqnhip_update_group-style grouping (not built: two streams deliver more)"}
        # configs[0]'s workload (the reference's compile-time defaults: minibatch 32, S = 59, tower 1024-512-256-128,
        # src/dqn.hpp:19, src/dqn.cpp:425) — on the GPU, since there is no CPU backend here; async and the drop-in's blocking form
        d = pkg.DQN(59, minibatch=32, hidden=(1024, 512, 256, 128), memory=100000, seed=1, device=local_rank, use_graph=True)
        from synth import synth_replay
        d.add_transitions_arrays(*synth_replay(np.random.default_rng(3), 50000, 59))
        dt = timed_n(d, torch.cuda.synchronize, 1000, 104)
        ms_b = d.BenchmarkBlocking(1000, 100, seed=1, pipelined=2)
        out["configs0_ref_defaults_b32"] = {"ms_per_update": round(dt * 1e3, 4), "updates_per_s": round(1 / dt, 1),
                                            "blocking_ms_per_update": round(ms_b, 4), "blocking_updates_per_s": round(1e3 / ms_b, 1),
                                            "plan": d.update_plan(),
                                            "note": "launch-bound: every launch of the plan at its ~5-us floor"}
        d.read_stats(); d.close()
        # configs[2]: 1v1 (S = 68), 64 parallel workers feeding one replay buffer
        d = pkg.DQN(68, minibatch=B, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, use_graph=True)
        env = pkg.EnvFrontEnd(d, 64, max_steps=500, p_end=0.01, seed=5)
        env.step(0.1, 20); env.stats()
        t1 = time.perf_counter(); env.step(0.1, 200); env.stats(); dt = (time.perf_counter() - t1) / 200
        out["configs2_64workers_s68"] = {"env_steps_per_s": round(64 / dt, 1), "us_per_batched_step": round(dt * 1e6, 2),
                                         "roofline": env_roofline(68, 64, dt * 1e6)}
        env.close(); d.close()
        # BOTH halves of BASELINE's metric at once: the reference's loop interleaves acting and learning — an
        # episode's steps, then `steps x update_ratio` updates (src/dqn_main.cpp:346-363, FLAGS_update_ratio 0.1) — on one thread per
        # agent.  Here: `workers` batched env steps then the updates they owe, enqueued back to back on the learner's stream (the env
        # front-end acts with the learner's actor and appends to its replay ring: the two are ordered, as in the reference), steady state.
        tl = {}
        for name, S_, workers, steps_per_burst in (("configs1_1_worker", S, 1, 160), ("configs2_64_workers_s68", 68, 64, 5), ("configs4_2048_workers", S, 2048, 1)):
            d = pkg.DQN(S_, minibatch=B, hidden=HIDDEN, memory=400000, seed=1, device=local_rank, use_graph=True)
            prefill(d, 100000, seed=7) if S_ == S else d.add_transitions_arrays(*__import__("synth").synth_replay(np.random.default_rng(3), 60000, S_))
            env = pkg.EnvFrontEnd(d, workers, max_steps=100 if workers >= 1024 else 500, p_end=0.01, seed=5)
            owed, bursts = 0.0, (40 if workers < 2048 else 12)
            def burst():
                nonlocal owed
                env.step(0.1, steps_per_burst)
                owed += workers * steps_per_burst * 0.1
                k = int(owed); owed -= k
                if k:
                    d.update_async_n(k)
                return k
            for _ in range(3):
                burst()
            env.stats(); d.read_stats()
            t1 = time.perf_counter(); n_up = sum(burst() for _ in range(bursts)); env.stats(); d.read_stats()
            dt = time.perf_counter() - t1
            # the same two loads alone, same learner, same process
            t1 = time.perf_counter(); env.step(0.1, steps_per_burst * bursts); env.stats(); dt_env = time.perf_counter() - t1
            d.update_async_n(64); d.read_stats()
            t1 = time.perf_counter(); d.update_async_n(max(n_up, 64)); d.read_stats(); dt_up = time.perf_counter() - t1
            tl[name] = {"workers": workers, "state_size": S_, "update_ratio": 0.1, "env_steps_per_burst": workers * steps_per_burst,
                        "updates_per_s": round(n_up / dt, 1), "env_steps_per_s": round(workers * steps_per_burst * bursts / dt, 1),
                        "isolated": {"env_steps_per_s": round(workers * steps_per_burst * bursts / dt_env, 1), "updates_per_s": round(max(n_up, 64) / dt_up, 1)},
                        "time_share_model": round((workers * steps_per_burst * bursts / dt) / (workers * steps_per_burst * bursts / dt_env) + (n_up / dt) / (max(n_up, 64) / dt_up), 3)}
            env.close(); d.close()
        tl["what"] = ("env steps and the updates they owe (update_ratio 0.1, src/dqn_main.cpp:346-363) enqueued alternately on the learner's stream; "
                      "time_share_model = concurrent / isolated rate, summed over the two loads (1.0 = the two simply share the stream's time)")
        out["train_loop"] = tl
        # what ONE rank of configs[4] on 8 GPUs runs (4096 / 8 = 512 rows), with the communicator in place (one rank:
        # the all-reduce moves nothing): the captured data-parallel update against the eager one and against the plain
        # captured update -> what the collectives' launches and the bf16 conversion cost beside the kernels
        rec = {}
        for prec, half in (("fp16", True), ("fp32", False)):
            r = {}
            forms = [("plain_graph", dict(use_graph=True), None), ("dp_eager", dict(use_graph=False), dict(half_grads=half)),
                     ("dp_graph", dict(use_graph=True), dict(half_grads=half))]
            if not half:
                forms.append(("dp_per_layer_graph", dict(use_graph=True), dict(per_layer=True)))
            # the sharded optimiser (DQNHIP_DP_SHARD_OPT): with ONE rank its slice is the whole arena, so this measures what
            # its extra launches cost (reduce-scatter + 4-float all-reduce + all-gathers instead of one all-reduce per net),
            # not what a 1/N slice saves — the projection takes that from profiles/r04_adam_slice_probe.txt
            forms.append(("dp_shard_graph", dict(use_graph=True), dict(half_grads=half, shard_opt=True)))
            for name, kw, dp_kw in forms:
                d = pkg.DQN(S, minibatch=512, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, precision=prec, **kw)
                prefill(d, 150000, seed=7)
                if dp_kw is not None:
                    d.dp_init(pkg.DQN.dp_unique_id(), **dp_kw)
                    step = lambda d=d: d.dp_update(None)
                else:
                    step = lambda d=d: d.update_async(None)
                r[name + "_ms"] = round(timed(step, torch.cuda.synchronize, 300, 30) * 1e3, 4)
                if dp_kw is not None and kw["use_graph"]:
                    r[name + "_captured"] = d.dp_graph_active()
                if name in ("plain_graph", "dp_graph"):
                    # the form the headline enqueues: sixteen updates per graph launch (dqnhip_update_async_n / dqnhip_dp_update_n), the
                    # next update's gather (and, where the riders fit, its first layers) riding in this update's optimiser launches
                    step_n = (lambda k, d=d: d.dp_update_n(k)) if dp_kw is not None else (lambda k, d=d: d.update_async_n(k))
                    step_n(64); torch.cuda.synchronize()
                    t1 = time.perf_counter(); step_n(320); torch.cuda.synchronize()
                    r[name + "_n_ms"] = round((time.perf_counter() - t1) / 320 * 1e3, 4)
                    r[name + "_plan"] = d.update_plan()
                if name == "plain_graph":
                    # the backward chain's launch durations at this rank shape: when each per-layer bucket becomes ready
                    d.set_kernel_timing(True)
                    for _ in range(10):
                        d.update_async(None)
                    fam = {f: d.kernel_timing(f)[0] * 1e3 for f in (("hgemm_dgrad", "hgemm_wgrad") if half else ("gemm_bwd_pair", "gemm_wgrad"))}
                    d.kernel_timing("adam", reset=True); d.set_kernel_timing(False)
                    r["backward_launch_us"] = {k: round(v, 2) for k, v in fam.items()}
                d.read_stats(); d.close()
            # the rank's compute at 4096 / N rows for N = 2 and 4 (plain captured update; the data-parallel launches add what
            # they add at 512 rows: dp_graph_ms - plain_graph_ms)
            for rows in (1024, 2048):
                d = pkg.DQN(S, minibatch=rows, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, precision=prec, use_graph=True)
                prefill(d, 150000, seed=7)
                r["plain_graph_ms_rows_%d" % rows] = round(timed(lambda d=d: d.update_async(None), torch.cuda.synchronize, 150, 20) * 1e3, 4)
                d.read_stats(); d.close()
            gb = grad_bytes(S, HIDDEN, half)
            one = out["configs4_1gpu_b4096_%s" % prec]["ms_per_update"]
            per = 2 if half else 4
            wa, wc = tower_weights(S, HIDDEN), tower_weights(S + 10, HIDDEN)
            slices = [[(w + HIDDEN[i]) * per for i, w in reversed(list(enumerate(ws)))] + [(nh * HIDDEN[-1] + nh) * per + 16]
                      for ws, nh in ((wc, 1), (wa, 10))]          # critic first (exchanged first), each: layer L-1 .. 0, head + tail
            bl = r["backward_launch_us"]
            def t_rank(n):
                base = {8: r["plain_graph_ms"], 4: r["plain_graph_ms_rows_1024"], 2: r["plain_graph_ms_rows_2048"]}[n]
                return {"single": base + r["dp_graph_ms"] - r["plain_graph_ms"], "sharded": base + r["dp_shard_graph_ms"] - r["plain_graph_ms"],
                        "per_layer": base + r.get("dp_per_layer_graph_ms", r["dp_graph_ms"]) - r["plain_graph_ms"]}
            r["exchange"] = "bf16 gradients + fp32 tails, one collective per net" if half else "fp32, one collective per net (per-layer buckets: dp_per_layer_graph_ms)"
            r["projection"] = {"n_gpus_%d" % n: dp_projection(n, half, gb, t_rank(n), one, bl.get("gemm_bwd_pair", bl.get("hgemm_dgrad", 0.0)),
                                                              bl.get("gemm_wgrad", bl.get("hgemm_wgrad", 0.0)), slices) for n in (2, 4, 8)}
            r["projection"]["ms_per_update_1gpu_b4096"] = one
            r["projection"]["speedup_without_collectives_8gpu"] = round(one / r["dp_graph_ms"], 2)
            if not half:
                # where north_star's ">= 6x strong scaling 1 -> 8" becomes reachable: the same model at a global minibatch of 16384
                # (2048 rows per rank: the collectives are the same bytes, the rank's compute 3x longer)
                d = pkg.DQN(S, minibatch=16384, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, precision=prec, use_graph=True)
                prefill(d, 150000, seed=7)
                one16 = timed(lambda d=d: d.update_async(None), torch.cuda.synchronize, 20, 3) * 1e3
                d.read_stats(); d.close()
                t8 = {k: r["plain_graph_ms_rows_2048"] + r["dp_graph_ms"] - r["plain_graph_ms"] for k in ("single", "per_layer", "sharded")}
                p16 = dp_projection(8, half, gb, t8, one16, bl.get("gemm_bwd_pair", 0.0), bl.get("gemm_wgrad", 0.0), slices)["single"]
                r["projection"]["global_minibatch_16384_n_gpus_8"] = {"ms_per_update_1gpu": round(one16, 3), "rank_rows": 2048, **p16}
            r["projection"]["model"] = ("PROJECTION for a global minibatch of 4096 = N x (4096 / N) rows; rank time = the plain captured update measured at 4096 / N rows "
                                        "+ what the data-parallel launches add at 512 rows; the backward launch durations that set the bucket-ready times are "
                                        "the 512-row ones; xGMI %.1f GB/s one way per link, %.0f us per collective; single = one all-reduce per net, "
                                        "nothing overlapped; per_layer = buckets on a communication stream, overlap with the rest of the backward chain credited "
                                        "(exposed_exchange_us); sharded = reduce-scatter + 4-float all-reduce + Adam on 1/N (profiles/r04_adam_slice_probe.txt) + "
                                        "all-gather of online AND target weights.  The first real multi-GPU run checks it."
                                        % (XGMI_LINK_GBS_ONE_WAY, COLLECTIVE_LATENCY_US))
            rec[prec] = r
        out["configs4_rank_shape_b512"] = rec
        if not (args.test_strong_record and native):
            return out
    if not native:
        return None
    # N > 1: env-steps/sec of the WHOLE job (the other half of BASELINE.json's metric): every rank steps its own workers
    # into its own replay shard — no collective — so the aggregate is ranks x workers x steps over the slowest rank's time
    # (barriers on both sides).  64 workers per GPU (configs[2]'s count) and 2048 / 8 = 256 (configs[4] on a full node).
    agg = {}
    d = pkg.DQN(S, minibatch=256, hidden=HIDDEN, memory=200000, seed=1 + rank, device=local_rank, use_graph=True)
    for workers in (64, 256):
        env = pkg.EnvFrontEnd(d, workers, max_steps=500, p_end=0.01, seed=5 + rank)
        env.step(0.1, 20); env.stats()
        n_env = 200

        def run():
            env.step(0.1, n_env); env.stats()
        t_env = timed(run, barrier, 1, 0)
        if rank == 0:
            agg["workers_%d_per_gpu" % workers] = {"n_gpus": world, "env_steps_per_s": round(world * workers * n_env / t_env, 1),
                                                   "us_per_batched_step": round(t_env / n_env * 1e6, 2),
                                                   "roofline_per_gpu": env_roofline(S, workers, t_env / n_env * 1e6)}
        env.close()
    d.close()
    if rank == 0:
        out["env_steps_all_ranks"] = agg
    # N > 1: STRONG scaling of a global minibatch of 4096 (where the batch warrants DP)
    for prec in ("fp32", "fp16"):
        rows = GB // world
        if rows % (128 if prec == "fp16" else 32):
            continue
        half = prec == "fp16" and not args.dp_fp32_grads
        d, dp = par.make_native_data_parallel(pkg, S, rank, world, local_rank, minibatch=rows, hidden=HIDDEN, memory=200000,
                                              seed=1, precision=prec, per_layer=args.dp_per_layer and prec == "fp32", half_grads=half,
                                              use_graph=not args.no_graph)
        prefill(d, 150000, seed=7 + rank)
        t_dp = timed(lambda: dp.update(None), barrier, n_it, 20)

        d.close()
        # the same rows without the two collectives (timing only): a learner of this rank's shape with no communicator,
        # driven phase by phase (the bf16-exchange learner itself refuses the phase API: its exchange lives in dp_update)
        loc = pkg.DQN(S, minibatch=rows, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, dp_world=max(world, 2), dp_rank=rank % max(world, 2),
                      precision=prec, use_graph=False)
        prefill(loc, 150000, seed=7 + rank)

        def local_only():
            loc.update_phase(0); loc.update_phase(1); loc.update_phase(2)
        t_loc = timed(local_only, barrier, n_it, 10)
        loc.close()
        t_one = None
        if rank == 0:                     # the single-GPU reference for the speed-up: whole global minibatch on rank 0
            one = pkg.DQN(S, minibatch=GB, hidden=HIDDEN, memory=200000, seed=1, device=local_rank, use_graph=True, precision=prec)
            prefill(one, 150000, seed=7)
            t_one = timed(lambda: one.update_async(None), torch.cuda.synchronize, n_it, 20)
            one.close()
        barrier()
        if rank == 0:
            gb = grad_bytes(S, HIDDEN, half)
            pr = [allreduce_projection_us(b, world) for b in gb]
            out["strong_b4096_%s" % prec] = {"n_gpus": world, "rows_per_gpu": rows, "ms_per_update": round(t_dp * 1e3, 4),
                                             "ms_per_update_without_collectives": round(t_loc * 1e3, 4),
                                             "allreduce_us_per_update": round((t_dp - t_loc) * 1e6, 1),
                                             "ms_per_update_1gpu": round(t_one * 1e3, 4), "speedup_vs_1gpu": round(t_one / t_dp, 3),
                                             "exchange": "bf16 gradients + fp32 tails" if half else "fp32",
                                             "projection": {"allreduce_bytes": gb, "allreduce_us_one_ring": sum(p["one_ring_us"] for p in pr),
                                                            "allreduce_us_all_links": sum(p["all_links_us"] for p in pr),
                                                            "speedup_one_ring": round(t_one / (t_loc + sum(p["one_ring_us"] for p in pr) * 1e-6), 3),
                                                            "speedup_all_links": round(t_one / (t_loc + sum(p["all_links_us"] for p in pr) * 1e-6), 3)}}
    return out if rank == 0 else None


def main():
    global B
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--replay", type=int, default=REPLAY)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--one-update-per-launch", action="store_true",
                    help="N = 1: enqueue the K steps as K calls of dqnhip_update_async (one hipGraph launch each) instead of one dqnhip_update_async_n(K)")
    ap.add_argument("--prewarm-ms", type=float, default=100.0, help="untimed load before the W warm-up steps (clock ramp, graph capture); 0: one update")
    ap.add_argument("--minibatch", type=int, default=B, help="rows per GPU (BASELINE metric: 256)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp16: tower GEMMs on fp16 MFMA with fp32 accumulate (BASELINE config #5); not the headline metric")
    ap.add_argument("--no-env", action="store_true", help="skip the env-steps/sec leg")
    ap.add_argument("--frames-per-trial", type=int, default=500)
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel path even with one rank (testing)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --minibatch is the GLOBAL minibatch, split over the ranks (value = global updates/s)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only for the 2-ranks-on-one-GPU flow test)")
    ap.add_argument("--share-device0", action="store_true",
                    help="testing: every rank uses GPU 0 (needs --backend gloo; RCCL refuses duplicate devices)")
    ap.add_argument("--dp-per-layer", action="store_true",
                    help="native DP, fp32 learner: per-layer buckets on a communication stream instead of ONE all-reduce per net (the projection "
                         "of sub_records.configs4_rank_shape_b512 prices both: with ~20 us per collective five buckets cost more than they hide)")
    ap.add_argument("--dp-shard-opt", action="store_true", help="native DP: sharded optimiser (DQNHIP_DP_SHARD_OPT) instead of the replicated one")
    ap.add_argument("--test-dp-probe", action="store_true", help="testing: run the captured-update probe with the ranks there are (N = 1 under --force-dp)")
    ap.add_argument("--test-dp-probe-fail", action="store_true", help="testing: pretend the probe failed (exercises the agreed fall-back to the eager update)")
    ap.add_argument("--test-dp-probe-multi-fail", action="store_true", help="testing: pretend only the sixteen-update graph failed the probe (fall-back: one update per graph launch)")
    ap.add_argument("--no-dp-probe", action="store_true",
                    help="N > 1: skip the sacrificial child group that tries the captured data-parallel update first (tests/dp_native_worker.py)")
    ap.add_argument("--dp-fp32-grads", action="store_true", help="native DP, fp16 learner: all-reduce fp32 gradients instead of bf16 (DQNHIP_DP_HALF_GRADS)")
    ap.add_argument("--dp-timeout", type=int, default=600, help="N > 1: seconds the headline measurement may take before every rank gives up")
    ap.add_argument("--tuning", type=int, default=0, help="dqnhip_config.tuning_flags (A/B switches, include/dqnhip.h DQNHIP_TUNE_*)")
    ap.add_argument("--test-strong-record", action="store_true", help="testing: also run the N > 1 strong-scaling record with the ranks there are")
    ap.add_argument("--trace-child", action="store_true", help="internal (live_kernel_trace): leave right after the K timed steps, so that they are the last updates in the kernel trace")
    ap.add_argument("--no-live-trace", action="store_true", help="roofline durations from eager HIP events around single launches instead of a live rocprofv3 --kernel-trace child pass (graph replay)")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC summary instead of two live rocprofv3 --pmc child passes")
    ap.add_argument("--no-subrecords", action="store_true", help="skip the per-config sub-records (configs #3, #5; strong scaling under N > 1)")
    ap.add_argument("--mode", default="dp", choices=["dp", "replicas"],
                    help="N>1: dp = gradient all-reduce (weak scaling), replicas = independent learners")
    args = ap.parse_args()
    if args.gpus > 1 and (args.dp_per_layer or args.dp_shard_opt):
        # the per-layer and sharded exchange forms have never run on more than one rank (dqnhip_dp_init refuses them for real
        # groups); the N-GPU line is measured on the one supported form: replicated optimiser, ONE all-reduce per net
        print("bench: --dp-per-layer / --dp-shard-opt are unverified on real links and ignored under --gpus > 1 (one bucket per net, replicated optimiser)", file=sys.stderr, flush=True)
        args.dp_per_layer = args.dp_shard_opt = False
    B = args.minibatch
    if args.strong:
        w_ = int(os.environ.get("WORLD_SIZE", "1"))
        if B % (32 * w_):
            raise SystemExit("--strong: global minibatch %d is not a multiple of 32 x %d ranks" % (B, w_))
        B //= w_

    import torch
    pkg = load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if args.share_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.force_dp
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    use_dp = dist_on and args.mode == "dp"
    native = use_dp and args.backend == "nccl"     # RCCL inside libdqnhip.so; gloo: torch.distributed between the phases
    par = None
    native_error = None
    if use_dp:
        from importlib import import_module
        par = import_module("dqn_hfo_amd.parallel")
        # ONE seed for the whole group: identical initialisation (rank 0's state is broadcast anyway), the
        # per-shard sample streams are decorrelated by dp_rank inside the library
        common = dict(minibatch=B, hidden=HIDDEN, memory=args.replay, seed=1, precision=args.precision, tuning=args.tuning)
        half = args.precision == "fp16" and not args.dp_fp32_grads
        dp_probe = None
        if native and (world > 1 or args.test_dp_probe) and not args.no_graph and not args.no_dp_probe:
            # The captured data-parallel update (RCCL kernels inside a hipGraph) has never met more than ONE rank on this repo's
            # build boxes, and a rank stuck in a collective cannot be interrupted from Python.  So a sacrificial child group
            # (one child per rank, own communicator, file rendezvous, no torch) tries it first at this shape; if any child
            # fails or has to be killed, every rank agrees to run the headline eagerly (same kernels, same collectives,
            # stream-ordered instead of replayed).  ~20 s, outside every timed region.
            dp_probe = (probe_captured_dp(rank, world, local_rank, args.precision, half, args.dp_per_layer and not half) if not args.test_dp_probe_fail
                        else {"ok": False, "why": "forced by --test-dp-probe-fail"})
            if args.test_dp_probe_multi_fail:
                dp_probe["multi_ok"] = False; dp_probe["why"] = "forced by --test-dp-probe-multi-fail"
            ok = torch.tensor([1 if dp_probe.get("ok") else 0, 1 if dp_probe.get("multi_ok") else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0].item()) == 0:
                args.no_graph = True
                if rank == 0:
                    print("bench: the captured data-parallel update did not pass the %d-rank probe (%s): running eagerly" % (world, dp_probe.get("why", "another rank failed")),
                          file=sys.stderr, flush=True)
            elif int(ok[1].item()) == 0:
                args.one_update_per_launch = True
                if rank == 0:
                    print("bench: the sixteen-update data-parallel graph did not pass the %d-rank probe (%s): one update per graph launch" % (world, dp_probe.get("why", "another rank failed")),
                          file=sys.stderr, flush=True)
        if native:
            # the communicator inside libdqnhip.so has never met more than one real GPU in this repo's own runs
            # (one-GPU boxes only): if ANY rank fails to bring it up, every rank falls back — by agreement over
            # torch.distributed — to the other transport of the same algorithm (torch's RCCL all-reduce between the
            # update phases), and the JSON line says so.  Same kernels, same numbers, a few host round trips more.
            try:
                dqn, dp = par.make_native_data_parallel(pkg, S, rank, world, local_rank, per_layer=args.dp_per_layer and not half and not args.dp_shard_opt,
                                                        half_grads=half, shard_opt=args.dp_shard_opt, use_graph=not args.no_graph, **common)
            except Exception as e:                                  # noqa: BLE001 — reported, not swallowed
                native_error = repr(e)[:300]
                dqn = dp = None
            if world > 1:
                ok = torch.tensor([0 if native_error else 1], dtype=torch.int32, device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    native_error = native_error or "another rank failed to initialise the native communicator"
            if native_error:
                if rank == 0:
                    print("bench: native RCCL data-parallel init failed (%s); falling back to torch.distributed all-reduce" % native_error,
                          file=sys.stderr, flush=True)
                if world == 1:
                    raise SystemExit("native data-parallel init failed: " + native_error)
                if dqn is not None:
                    try:
                        dqn.close()
                    except Exception:                               # noqa: BLE001
                        pass
                    del dp, dqn
                native = False
        if not native:
            dqn, dp = par.make_hip_data_parallel(pkg, S, rank, world, local_rank, **common)
        step = lambda: dp.update(None)
    else:
        dqn = pkg.DQN(S, minibatch=B, hidden=HIDDEN, memory=args.replay, seed=1 + rank, device=local_rank,
                      use_graph=not args.no_graph, precision=args.precision, tuning=args.tuning)
        step = lambda: dqn.update_async(None)
    # K updates in a row: ONE call where the library has one (dqnhip_update_async_n — the reference's own inner loop
    # `for (i < n_updates) dqn->Update()`, src/dqn_main.cpp:359-361 / DQN::Benchmark, src/dqn.cpp:487-498 — which replays sixteen
    # updates per hipGraph launch), else K calls.  --one-update-per-launch: K calls of dqnhip_update_async also at N = 1.
    # Data parallel, native transport: dqnhip_dp_update_n, the same for a group.
    batched_enqueue = not args.one_update_per_launch and (not use_dp or native)

    def steps(k):
        if batched_enqueue and use_dp:
            dp.update_n(k)
        elif batched_enqueue:
            dqn.update_async_n(k)
        else:
            for _ in range(k):
                step()
    prefill(dqn, args.replay - 1, seed=100 + rank)     # AddTransitions keeps <= capacity-1 (src/dqn.cpp:776)

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the captured data-parallel update (RCCL inside a hipGraph, per-layer buckets on a communication stream) has
    # only ever met ONE rank on this repo's build boxes.  A rank stuck in a collective cannot be interrupted from Python, so a
    # watchdog bounds the damage: if the headline has not been measured within --dp-timeout seconds every rank leaves, and rank 0
    # prints a line that says so (value null) instead of hanging the driver.  Re-run with --no-graph / --dp-single-bucket.
    hang_dog = None
    if dist_on and world > 1:
        import threading

        def _hung():
            if rank == 0:
                print(json.dumps({"metric": "DQN updates/sec, 1v0 HFO, 4x1024 MLP, minibatch %d" % B, "value": None, "unit": "updates/s",
                                  "n_gpus": world, "error": "data-parallel headline did not finish within %d s (graph=%s, per_layer=%s): "
                                  "rerun with --no-graph" % (args.dp_timeout, not args.no_graph, args.dp_per_layer)}), flush=True)
            os._exit(3)
        hang_dog = threading.Timer(float(args.dp_timeout), _hung)
        hang_dog.daemon = True
        hang_dog.start()

    # Untimed preparation that is not one of the W warm-up steps: the first update captures the hipGraph, and
    # the GPU needs a few tens of milliseconds of load before its clocks settle.  With a short --warmup / --steps
    # pair (the driver has used 5 / 20 = 9 ms in total) the timed region would otherwise sit on that ramp.
    # Reported as config.prewarm_updates; the W warm-up steps and the K timed steps follow unchanged.
    prewarm = 0
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < args.prewarm_ms * 1e-3 or prewarm < 1:
        steps(16)
        prewarm += 16
        if dist_on:
            break                       # collectives: every rank must run the same count
    barrier()
    steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if args.trace_child:
        print(json.dumps({"trace_child": True, "ms_per_step": elapsed / args.steps * 1e3}), flush=True)
        dqn.close()
        return                               # (a normal exit: the profiler writes its trace from its exit handlers)
    one_per_launch_ms = None
    if batched_enqueue and not args.no_graph:
        # the same K updates as K calls of dqnhip_update_async (one update per hipGraph launch), for the record
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        one_per_launch_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if hang_dog is not None:
        hang_dog.cancel()
    if dist_on:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss, avgq = dqn.read_stats()
    final_line = None
    # N > 1: the one-GPU rate measured in THIS job (a plain learner per rank, no communicator, all ranks at once so that they
    # stay in step), so that a SCALE record can be cross-checked against the BENCH record of the same round without trusting
    # that two boxes clock alike
    n1_same_job = None
    if dist_on and world > 1 and not args.strong:
        d1 = pkg.DQN(S, minibatch=B, hidden=HIDDEN, memory=100000, seed=1, device=local_rank, use_graph=not args.no_graph,
                     precision=args.precision, tuning=args.tuning)
        prefill(d1, 60000, seed=5 + rank)
        dt1 = timed_n(d1, torch.cuda.synchronize, 320, 64)
        d1.read_stats(); d1.close()
        barrier()
        n1_same_job = {"updates_per_s": round(1 / dt1, 1), "ms_per_step": round(dt1 * 1e3, 5),
                       "what": "rank 0's plain single learner (dqnhip_update_async_n, 320 updates, 60 000-transition replay) while every other rank runs its own: the N = 1 point of the same job"}

    # roofline of the dominant kernel family, timed live with HIP events on the learner's stream
    roof = None
    n_t = 20
    # every rank runs the extra updates (under data parallelism they contain collectives); only rank 0
    # brackets its launches with events
    if rank == 0:
        dqn.set_kernel_timing(True)
    for _ in range(n_t):
        step()
    if rank == 0:
        fp16 = args.precision == "fp16"
        fam_flops = family_flops16(B, S, HIDDEN) if fp16 else family_flops(B, S, HIDDEN, shifted=not (args.tuning & 4) and B >= 64)
        peak = MFMA_F16_PEAK_TF if fp16 else MFMA_F32_PEAK_TF
        stats = {}
        for fam in list(fam_flops) + ["adam"]:
            ms, cnt = dqn.kernel_timing(fam)
            stats[fam] = (ms, cnt)
        dqn.kernel_timing("adam", reset=True)
        dqn.set_kernel_timing(False)
        events_stats = dict(stats)
        dur_src = "HIP events (hipExtLaunchKernelGGL start / stop) around each launch of %d eager updates after the timed region" % n_t
        trace = None
        extra = ["--minibatch", str(B), "--precision", args.precision, "--tuning", str(args.tuning)]
        if world == 1 and not args.no_live_trace and not args.no_graph:
            trace, why = live_kernel_trace(extra)
            if trace is None:
                dur_src += " [live kernel trace unavailable: %s]" % why
            else:
                # fold the traced kernels into the timing families: (mean ms per launch, launches in n_t updates)
                fam_acc = {}
                for kname, (us, per_upd) in trace.items():
                    fam = next((f for pre, f in TRACE_FAMILY if kname.startswith(pre)), None)
                    if fam is None:
                        continue
                    a = fam_acc.setdefault(fam, [0.0, 0.0])
                    a[0] += us * per_upd; a[1] += per_upd
                if all(f in fam_acc for f in fam_flops):
                    stats = {f: ((fam_acc[f][0] / fam_acc[f][1]) * 1e-3, fam_acc[f][1] * n_t) for f in fam_acc}
                    dur_src = why
                else:
                    dur_src += " [live kernel trace lacks families: %s]" % sorted(set(fam_flops) - set(fam_acc))
                    trace = None
        dom = max(fam_flops, key=lambda f: stats[f][0] * stats[f][1])
        ms, cnt = stats[dom]
        per_update_launches = cnt / n_t
        flops_per_launch = fam_flops[dom] / per_update_launches
        ach = flops_per_launch / (ms * 1e-3) / 1e12
        traffic, traffic_src = (None, None)
        if world == 1 and not args.no_live_pmc:
            traffic, traffic_src = live_pmc_traffic(KERNEL_NAMES[dom], extra)
        if traffic is None:
            why = traffic_src
            traffic, traffic_src = pmc_traffic(KERNEL_NAMES[dom])
            if traffic_src and why:
                traffic_src += " [live PMC pass unavailable: %s]" % why
        roof = {"bound": "mfma", "kernel": KERNEL_NAMES[dom], "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_us": round(ms * 1e3, 2), "launches_per_update": per_update_launches,
                "flops_per_launch": flops_per_launch,
                "duration_source": dur_src,
                "families_us": {f: [round(stats[f][0] * 1e3, 2), round(stats[f][1] / n_t, 3)] for f in stats},
                # the same families timed with eager HIP events in this process (a cross-check; differs by a few per cent run to run)
                "families_us_eager_events": {f: [round(events_stats[f][0] * 1e3, 2), events_stats[f][1] / n_t] for f in events_stats},
                **({"kernel_trace_us": {k: [round(v[0], 2), round(v[1], 3)] for k, v in sorted(trace.items()) if v[1] >= 0.5}} if trace else {}),
                # every GEMM family's own fraction of the peak (algorithmic FLOPs of its launches / their summed duration)
                "families_frac": {f: round(fam_flops[f] * n_t / (stats[f][0] * 1e-3 * stats[f][1]) / 1e12 / peak, 4) for f in fam_flops if stats[f][1] / n_t >= 0.5}}     # (a family whose work rides elsewhere in most updates - the first layers inside a graph - has no launch of its own to rate)
    torch.cuda.synchronize()

    # What the reference's UNCHANGED driver gets through the drop-in (src/dqn_main.cpp:361 -> DQN::Update ->
    # UpdateActorCritic; DQN::Benchmark, src/dqn.cpp:487-498): indices drawn on the host (std::mt19937), staged H2D,
    # and a blocking (loss, avg_q) read-back per update — against the async headline above.  Measured inside the
    # library (dqnhip_benchmark_blocking), so no Python is in the loop; and the one-deep pipelined read-back.
    dropin = None
    if rank == 0 and not dist_on and not args.no_subrecords:
        ms_b = dqn.BenchmarkBlocking(1000, 100, seed=1, pipelined=False)
        ms_p = dqn.BenchmarkBlocking(1000, 100, seed=1, pipelined=True)
        ms_c = dqn.BenchmarkBlocking(1000, 100, seed=1, pipelined=2)
        ms_a = elapsed / args.steps * 1e3
        dropin = {"blocking_ms_per_update": round(ms_c, 5), "blocking_updates_per_s": round(1e3 / ms_c, 1),
                  "blocking_form": "dqnhip_update_chained: what the drop-in's UpdateActorCritic() calls since round 6 (the next call's indices predicted "
                                   "from a copy of the driver's std::mt19937; the next gather / first layers ride in this update's optimiser launches)",
                  "unchained_blocking_ms_per_update": round(ms_b, 5), "unchained_blocking_updates_per_s": round(1e3 / ms_b, 1),
                  "pipelined_ms_per_update": round(ms_p, 5), "pipelined_updates_per_s": round(1e3 / ms_p, 1),
                  "async_headline_ms_per_update": round(ms_a, 5),
                  "blocking_vs_async": round(ms_c / ms_a, 4), "unchained_blocking_vs_async": round(ms_b / ms_a, 4), "pipelined_vs_async": round(ms_p / ms_a, 4),
                  "what": "1000 blocking updates with host-drawn indices and a (loss, avg_q) read-back each = DQN::UpdateActorCritic() of the "
                          "drop-in (dqn_dropin.cpp), i.e. DQN::Benchmark / dqn_main.cpp:361; unchained = plain dqnhip_update (-chained_updates=false, "
                          "rounds 3-5); pipelined = dqnhip_update_pipelined "
                          "(-pipelined_stats: returns the previous update's scalars)"}

    # env-steps/sec (the other half of BASELINE.json's metric): N synthetic workers -> batched
    # SelectActions + GetAction + HFOGameState reward + LabelTransitions/AddTransitions, all on device
    env_res = None
    if rank == 0 and not args.no_env:
        env_res = {}
        for workers in (64, 256, 1024, 2048):     # BASELINE configs #3 (64 workers) and #5 (2048 workers; 256 = its share per GPU on 8)
            # episode buffers must fit the replay ring: cap the episode length for the widest runs
            T = min(args.frames_per_trial, (args.replay - 1) // workers)
            if T < 100:
                continue
            env = pkg.EnvFrontEnd(dqn, workers, max_steps=T, p_end=0.01, seed=5)
            env.step(0.1, 20)
            env.stats()
            t1 = time.perf_counter()
            n_env = 200 if workers <= 64 else 100
            env.step(0.1, n_env)
            st = env.stats()          # blocks
            dt = time.perf_counter() - t1
            env_res["workers_%d" % workers] = {"env_steps_per_s": round(workers * n_env / dt, 1),
                                               "us_per_batched_step": round(dt / n_env * 1e6, 2),
                                               "episodes": st[1], "max_episode_steps": T,
                                               "roofline": env_roofline(S, workers, dt / n_env * 1e6)}
            env.close()
        if not args.no_cpu_baseline and world == 1:
            from oracle import c_oracle, torch_ref
            orc = c_oracle.Oracle(B=B, S=S, hidden=HIDDEN, capacity=200000)
            c_oracle.set_threads(c_oracle.usable_cores())
            rngw = np.random.default_rng(3)
            orc.set_params(0, torch_ref.init_params_np(rngw, S, HIDDEN, True))
            oenv = c_oracle.OracleEnv(orc, 64, max_steps=args.frames_per_trial, p_end=0.01, seed=5)
            oenv.step(0.1, 1)
            t1 = time.perf_counter(); oenv.step(0.1, 10); dt = time.perf_counter() - t1
            env_res["cpu_port_workers_64"] = {"env_steps_per_s": round(64 * 10 / dt, 1), "cores": c_oracle.usable_cores()}
            oenv.close(); orc.close()

    out = None
    if rank == 0:
        try:
            plan = dqn.update_plan()
        except Exception as ex:        # (a sharded optimiser's sequence is not counted; never fatal for the line)
            plan = {"error": str(ex)}
        ups = args.steps / elapsed
        value = ups if args.strong else ups * (world if world > 1 else 1)
        fl = sum(family_flops(B, S, HIDDEN).values())
        out = {
            "metric": "DQN updates/sec, 1v0 HFO, 4x1024 MLP, minibatch %d" % (B * world if args.strong else B),
            "value": round(value, 2), "unit": "updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16 operands, f32 accumulate (master weights / Adam / heads f32)",
            "data": "synthetic",
            # (the driver keeps 120 characters of `workload`: the sizes it cut off last round are keys of their own)
            "config": {"workload": "BASELINE configs[1]: 1v0 HFO, 1 MI355X, 4x1024 actor-critic, minibatch %d, %s-transition device replay" % (B, ("%dM" % (args.replay // 1000000)) if args.replay % 1000000 == 0 else str(args.replay)),
                       "replay_transitions": args.replay, "state_size": S, "hidden": list(HIDDEN), "minibatch": B,
                       "minibatch_per_gpu": B, "global_minibatch": B * (world if use_dp else 1),
                       "parallelism": ("dp%d (%s all-reduce of critic then actor gradients); value counts %s"
                                       % (world, "RCCL inside libdqnhip.so (dqnhip_dp_update)" if native else args.backend + " via torch.distributed",
                                          "global-minibatch updates" if args.strong else "minibatch-%d updates" % B)) if use_dp else
                                      ("replicas x%d" % world if world > 1 else "single"),
                       "hip_graph": (not args.no_graph) and (not use_dp or (native and dqn.dp_graph_active())), "prewarm_updates": prewarm,
                       "enqueue": ("%s(K): K updates in one call, replayed sixteen per hipGraph launch, each update's gather and first tower layers riding in the previous update's optimiser launches"
                                   % ("dqnhip_dp_update_n" if use_dp else "dqnhip_update_async_n") if batched_enqueue
                                   else "one call (one hipGraph launch) per update"),
                       **({"ms_per_step_one_update_per_graph_launch": round(one_per_launch_ms, 5)} if one_per_launch_ms else {}),
                       "tuning_flags": args.tuning,
                       "plan": plan,        # dqnhip_get_update_plan: the merged launch forms this learner runs + kernels per update, counted from a capture
                       **({"n1_same_job": n1_same_job} if n1_same_job else {}),
                       **({"native_dp_error": native_error} if native_error else {}),
                       **({"captured_dp_probe": dp_probe} if use_dp and dp_probe is not None else {}),
                       "sampling": "on-device Philox, uniform with replacement"},
            "update_gflop": round(fl / 1e9, 3),
            "update_mfma_frac": round(fl * ups / 1e12 / (MFMA_F16_PEAK_TF if args.precision == "fp16" else MFMA_F32_PEAK_TF), 4),
            "last_critic_loss": loss, "last_avg_q": avgq,
            "roofline": roof,
            "env_steps": env_res,
            "sub_records": None,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, S, HIDDEN)
        else:
            out["cpu_baseline"] = None

    def emit_and_exit(note=None):
        """The ONE JSON line (rank 0), then leave without running the communication libraries' exit handlers."""
        if rank == 0:
            if note:
                out["sub_records"] = {"error": note}
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # RCCL's banner goes through C stdio: push it out first
            except Exception:
                pass
            sys.stdout.flush()
            print(json.dumps(out), flush=True)
        os._exit(0)

    # side records (configs[2], configs[4]; strong scaling under N > 1).  They must never cost the headline:
    # a watchdog on every rank prints the line without them if they do not finish (e.g. one rank stuck in a
    # collective the others never entered), and an exception only drops the side records.
    if not args.no_subrecords and not args.strong and args.precision == "fp32" and B == 256:
        import threading
        dog = threading.Timer(300.0, emit_and_exit, kwargs={"note": "side records did not finish within 300 s"})
        dog.daemon = True
        dog.start()
        try:
            sub = sub_records(pkg, par, args, rank, world, local_rank, native, barrier)
        except Exception as ex:
            sub = {"error": "%s: %s" % (type(ex).__name__, ex)}
        dog.cancel()
        if rank == 0:
            if isinstance(sub, dict) and dropin is not None:
                sub["dropin_blocking_b%d" % B] = dropin
            out["sub_records"] = sub
    if rank == 0:
        final_line = json.dumps(out)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    dqn.close()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which is flushed only at exit: push it out
        # now so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(final_line, flush=True)       # the ONE JSON line
        os._exit(0) if dist_on else None    # nothing after it (atexit output of the comm libraries)


if __name__ == "__main__":
    main()
