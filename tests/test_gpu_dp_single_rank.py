"""The data-parallel plumbing on one real GPU: world_size 1, backend nccl (= RCCL).  Exercises
what N > 1 uses — gradient arenas inside a torch tensor, the learner enqueuing on torch's
current stream, the three-phase update with all-reduces in between — and checks it against the
single-GPU path (a 1-rank sum all-reduce is the identity)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from __graft_entry__ import load_package
from importlib import import_module
pkg = load_package()
par = import_module("dqn_hfo_amd.parallel")
from oracle import torch_ref
from synth import synth_replay
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, S, hid = 64, 59, (256, 128, 64, 64)
rng = np.random.default_rng(5)
w = [torch_ref.init_params_np(rng, S, hid, a) * 5 for a in (True, False)]
data = synth_replay(rng, 1024, S, mean_len=10)
idx = rng.integers(0, 1024, size=(4, B))
dqn_dp, dp = par.make_hip_data_parallel(pkg, S, 0, 1, 0, minibatch=B, hidden=hid, memory=4096, seed=2)
ref = pkg.DQN(S, minibatch=B, hidden=hid, memory=4096, seed=2)
for d in (dqn_dp, ref):
    for net in (0, 1):
        d.set_params(net, w[net]); d.CloneNet(net)
    d.add_transitions_arrays(*data)
for u in range(4):
    dp.update(idx[u]); s_dp = dqn_dp.read_stats()
    s_ref = ref.UpdateActorCritic(idx[u])
    assert abs(s_dp[0] - s_ref[0]) <= 1e-5 * max(1, abs(s_ref[0])), (s_dp, s_ref)
    assert abs(s_dp[1] - s_ref[1]) <= 1e-5, (s_dp, s_ref)
for net in range(4):
    a, b = dqn_dp.get_params(net), ref.get_params(net)
    assert np.abs(a - b).max() <= 1e-6, (net, np.abs(a - b).max())
assert not dp.overlap
# the overlap form (phase 10 / async all-reduce / phase 11) gives the same bits (10 + 11 == 0)
dqn_b, dp_b = par.make_hip_data_parallel(pkg, S, 0, 1, 0, overlap=True, minibatch=B, hidden=hid, memory=4096, seed=2)
assert dp_b.overlap
for net in (0, 1):
    dqn_b.set_params(net, w[net]); dqn_b.CloneNet(net)
dqn_b.add_transitions_arrays(*data)
for u in range(4):
    dp_b.update(idx[u])
for net in range(4):
    np.testing.assert_array_equal(dqn_b.get_params(net), dqn_dp.get_params(net))
for u in range(3):                       # on-device sampling through the DP path
    dp.update(None)
l, q = dqn_dp.read_stats()
assert np.isfinite(l) and np.isfinite(q)
# the mixed-precision learner through the same three phases (bit-identical: same kernels, same order)
B16, hid16 = 128, (256, 128)
w16 = [torch_ref.init_params_np(rng, S, hid16, a) * 5 for a in (True, False)]
idx16 = rng.integers(0, 1024, size=(3, B16))
d16, dp16 = par.make_hip_data_parallel(pkg, S, 0, 1, 0, minibatch=B16, hidden=hid16, memory=4096, seed=2, precision="fp16")
r16 = pkg.DQN(S, minibatch=B16, hidden=hid16, memory=4096, seed=2, precision="fp16")
for d in (d16, r16):
    for net in (0, 1):
        d.set_params(net, w16[net]); d.CloneNet(net)
    d.add_transitions_arrays(*data)
for u in range(3):
    dp16.update(idx16[u]); r16.UpdateActorCritic(idx16[u])
for net in range(4):
    np.testing.assert_array_equal(d16.get_params(net), r16.get_params(net))
dist.barrier(); dist.destroy_process_group()
print("DP-1 OK")
'''


def test_dp_plumbing_world_size_one(pkg, gpu):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DP-1 OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_runs_under_torchrun_one_rank(pkg, gpu):
    """The exact launch line the driver uses for N > 1, with N = 1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "20", "--warmup", "5", "--replay", "20000", "--no-cpu-baseline", "--force-dp", "--test-dp-probe"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 100 and d["config"]["parallelism"].startswith("dp1")
    assert d["config"]["hip_graph"] is True                     # the data-parallel update (RCCL included) replayed as a graph
    # under N > 1 a sacrificial child group tries the captured update first (tests/dp_native_worker.py --mode probe); the same
    # code with the one rank there is: the child ran at the bench's shape, its graph was active and equal to the eager member's
    probe = d["config"]["captured_dp_probe"]
    assert probe["ok"] and probe["multi_ok"] and probe["graph_ms_per_update"] > 0 and probe["graph_n_ms_per_update"] > 0, probe
    assert d["config"]["enqueue"].startswith("dqnhip_dp_update_n")


def test_bench_falls_back_to_eager_when_the_probe_fails(pkg, gpu):
    """If the sacrificial child group does not get through the captured data-parallel update, every rank agrees to run the
    headline eagerly: same kernels, same collectives, a line with hip_graph false and the reason on record."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29649", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "20", "--warmup", "5", "--replay", "20000", "--no-cpu-baseline", "--no-env", "--no-subrecords", "--no-live-pmc",
           "--force-dp", "--test-dp-probe", "--test-dp-probe-fail"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] > 100 and d["config"]["hip_graph"] is False
    assert d["config"]["captured_dp_probe"]["ok"] is False and "running eagerly" in r.stderr


def test_bench_falls_back_to_one_update_per_graph_when_only_the_multi_update_graph_fails(pkg, gpu):
    """The probe child reports stage by stage: if the one-update graph is fine but dqnhip_dp_update_n's sixteen-update graph is
    not, the ranks keep replaying graphs, one update per launch."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29651", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "20", "--warmup", "5", "--replay", "20000", "--no-cpu-baseline", "--no-env", "--no-subrecords", "--no-live-pmc",
           "--force-dp", "--test-dp-probe", "--test-dp-probe-multi-fail"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] > 100 and d["config"]["hip_graph"] is True
    assert d["config"]["captured_dp_probe"]["ok"] is True and d["config"]["captured_dp_probe"]["multi_ok"] is False
    assert "one call (one hipGraph launch) per update" in d["config"]["enqueue"] and "one update per graph launch" in r.stderr


def test_bench_strong_scaling_record_with_the_ranks_there_are(pkg, gpu):
    """The N > 1 side record (a global minibatch of 4096 split over the ranks: captured native RCCL update, fp32 and fp16 with the bf16 exchange, the no-collective twin, the single-GPU reference, the projection) has only ever
    been reachable on a multi-GPU node; --test-strong-record runs the same code with one rank."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29647", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "20", "--warmup", "5", "--replay", "20000", "--no-cpu-baseline", "--no-env", "--force-dp",
           "--test-strong-record", "--no-live-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sub = d["sub_records"]
    assert "error" not in sub, sub
    for prec, exch in (("fp32", "fp32"), ("fp16", "bf16 gradients + fp32 tails")):
        rec = sub["strong_b4096_%s" % prec]
        assert rec["rows_per_gpu"] == 4096 and rec["exchange"] == exch
        assert 0 < rec["ms_per_update_without_collectives"] and 0 < rec["ms_per_update"] < 2 * rec["ms_per_update_1gpu"] + 1
        assert rec["projection"]["allreduce_us_one_ring"] == 0.0           # one rank: nothing crosses a link
    # whole-job env-steps/sec (every rank steps its own workers into its own replay shard)
    for workers in (64, 256):
        e = sub["env_steps_all_ranks"]["workers_%d_per_gpu" % workers]
        assert e["n_gpus"] == 1 and e["env_steps_per_s"] > 1e5 and 0 < e["roofline_per_gpu"]["frac"] < 1


def test_bench_two_ranks_flow_on_one_gpu(pkg, gpu):
    """The whole N = 2 bench flow (rendezvous, per-rank replay shards, the two all-reduces per update,
    the all-rank timing pass, rank-0-only legs, final barrier) with both ranks on GPU 0 and gloo as
    the transport — RCCL refuses two ranks on one device, but a collective that only some ranks
    enter hangs the same way on either backend."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29643", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "5", "--replay", "20000", "--no-cpu-baseline",
           "--backend", "gloo", "--share-device0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 10 and d["config"]["parallelism"].startswith("dp2")
    assert d["config"]["global_minibatch"] == 512 and d["roofline"] is not None
    # the N = 1 point measured in the same job (two plain learners share GPU 0 here, so only its presence and sanity are checked)
    n1 = d["config"]["n1_same_job"]
    assert n1["updates_per_s"] > 100 and abs(n1["ms_per_step"] * n1["updates_per_s"] - 1e3) < 1.0


def test_unverified_exchange_forms_are_fenced_for_real_groups(pkg, gpu):
    """DQNHIP_DP_PER_LAYER / DQNHIP_DP_SHARD_OPT have only ever run on one-rank groups: dqnhip_dp_init refuses them for
    dp_world > 1 unless DQNHIP_DP_UNVERIFIED_OK is passed — before any communicator is created, so a rank of a 2-rank group
    on this one-GPU box can show it without a peer.  One-rank groups (the launch-sequence tests) stay open."""
    d = pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=256, dp_world=2, dp_rank=0)
    uid = pkg.DQN.dp_unique_id()
    for kw in (dict(per_layer=True), dict(shard_opt=True), dict(shard_opt=True, half_grads=True)):
        with pytest.raises(pkg.DQNFatal, match="never run on more than one rank"):
            d.dp_init(uid, **kw)
    d.close()
    d1 = pkg.DQN(59, minibatch=32, hidden=(64, 64), memory=256)
    d1.dp_init(pkg.DQN.dp_unique_id(), shard_opt=True)          # world 1: allowed
    ver, path = pkg.DQN.dp_info()
    assert ver > 20000 and "rccl" in path, (ver, path)
    d1.dp_destroy(); d1.close()
