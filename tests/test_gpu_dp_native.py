"""The NATIVE data-parallel path (RCCL inside libdqnhip.so) with REAL ranks: one process per GPU, started the way a C++
host would start them (no torch, file rendezvous).  VERDICT r3 missing #1: every earlier RCCL test was a one-rank
communicator.

  * world = 1 always runs (any GPU box): the worker script, the rendezvous, the captured-graph-vs-eager comparison and
    the comparison with a plain learner — so that the day a multi-GPU lease appears the only new thing is the second rank.
  * world = 2 (and 4 / 8 when visible) runs when hipGetDeviceCount() >= 2 and is skipped, with that reason, otherwise:
    replicas bit-identical across ranks after every update, the group within float round-off of ONE learner on the
    concatenated minibatch (src/dqn.cpp:828-972 at the global batch), eager == captured hipGraph bit for bit, for the
    fp32 learner with per-layer buckets, the fp32 learner with one bucket, the fp16 learner with bf16 exchange, and the
    sharded-optimiser form.
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_native_worker.py")


def device_count():
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def run_group(tmp_path, world, extra, timeout=240, mode="parity"):
    """start `world` worker processes (rank r on device r), wait, return their result dicts"""
    rv = str(tmp_path / "rv")
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, "--rank", str(r), "--world", str(world), "--device", str(r), "--rv", rv,
                               "--out", outs[r], "--mode", mode] + list(extra),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    t0 = time.time()
    logs = []
    hung = False
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            hung = True
            p.kill()                      # exactly the process started above
            out, _ = p.communicate()
        logs.append(out[-3000:])
    assert not hung, "a rank did not finish within %d s (a collective that never completes?):\n%s" % (timeout, "\n---\n".join(logs))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r])
    res = [json.load(open(o)) for o in outs]
    assert all(x["ok"] for x in res), res
    return res


CONFIGS = [
    pytest.param(["--precision", "fp32", "--per-layer"], id="fp32-per-layer-buckets"),
    pytest.param(["--precision", "fp32"], id="fp32-one-bucket"),
    pytest.param(["--precision", "fp16", "--half", "--rows", "128", "--hidden", "256,128"], id="fp16-bf16-exchange"),
    pytest.param(["--precision", "fp32", "--shard-opt"], id="fp32-sharded-optimiser"),
    pytest.param(["--precision", "fp16", "--half", "--shard-opt", "--rows", "128", "--hidden", "256,128"], id="fp16-bf16-sharded-optimiser"),
]


def check(res, world, extra):
    half = "--half" in extra
    fp16 = "fp16" in extra
    for x in res:
        assert x["graph_active"], "RCCL refused the capture: the data-parallel update ran eagerly"
        assert x["graph_equals_eager"], "a graph-replaying group member and an eager one diverged"
        assert x["iters"] == [3, 3]
    # replicas: every rank holds the same bits (identical Adam step on the reduced gradient)
    assert len({x["eager_digest"] for x in res}) == 1, [x["eager_digest"] for x in res]
    assert len({x["graph_digest"] for x in res}) == 1
    assert all(x["eager_stats"] == res[0]["eager_stats"] for x in res)        # (critic_loss, avg_q): all-reduced tails
    v = res[0]["vs_single_learner"]
    # the group against ONE learner on the concatenated minibatch
    g_tol = 2e-2 if half else (2e-3 if fp16 else 1e-5)          # bf16 exchange: 8 significant bits; fp16: tile membership of rows
    w_max, w_mean = (3.5, 0.06) if half else ((3.0, 0.02) if fp16 else (3.0, 0.01))
    if not ("--shard-opt" in extra and world > 1):          # (sharded: a rank's arena holds the REDUCED gradient on its own slice only)
        assert v["g0"] <= g_tol and v["g1"] <= g_tol, v
    for net in range(4):
        mx, mean = v["w%d" % net]
        assert mx <= w_max + 0.1 and mean <= w_mean, (net, v)                # in units of one Adam step (lr)
    for (ls, qs), (lo, qo) in zip(res[0]["eager_stats"], res[0]["one_stats"]):
        tol = 5e-3 if (half or fp16) else 1e-5
        assert abs(ls - lo) <= tol * max(1.0, abs(lo)) and abs(qs - qo) <= tol * max(1.0, abs(qo)), (res[0]["eager_stats"], res[0]["one_stats"])


@pytest.mark.parametrize("extra", CONFIGS)
def test_native_dp_worker_one_rank(gpu, tmp_path, extra):
    res = run_group(tmp_path, 1, extra)
    check(res, 1, extra)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("extra", CONFIGS)
def test_native_dp_real_ranks(gpu, tmp_path, extra, world):
    n = device_count()
    if n < world:
        pytest.skip("needs %d GPUs, hipGetDeviceCount() = %d: runs the day a multi-GPU lease appears" % (world, n))
    res = run_group(tmp_path, world, extra)
    check(res, world, extra)
