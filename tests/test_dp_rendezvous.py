"""dqnhip_dp_rendezvous_file (include/dqnhip.h): the file rendezvous of a data-parallel group needs no GPU — rank 0
hands a 128-byte id to ranks 1..world-1 through files.  Checked here with one thread per rank (ctypes drops the
GIL): stale files of an earlier job, a waiter that starts before rank 0, reuse of the path, time-outs."""
import os
import threading
import time

import pytest


def _group(pkg, path, world, uid, delays, timeout_s=20):
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            time.sleep(delays[r])
            out[r] = pkg.dp_rendezvous_file(path, r, world, uid if r == 0 else None, timeout_s=timeout_s)
        except Exception as e:          # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join()
    return out, err


@pytest.mark.parametrize("delays", [(0.0, 0.0, 0.0, 0.0), (0.5, 0.0, 0.0, 0.2), (0.0, 0.4, 0.1, 0.0)])
def test_rendezvous_hands_rank0_id_to_every_rank(pkg, tmp_path, delays):
    path = str(tmp_path / "id")
    uid = bytes(range(128))
    out, err = _group(pkg, path, 4, uid, delays)
    assert err == [None] * 4, err
    assert all(o == uid for o in out)
    pkg.dp_rendezvous_cleanup(path, 4)
    assert not any(f.startswith("id") for f in os.listdir(tmp_path))


def test_stale_files_of_an_earlier_job_are_never_accepted(pkg, tmp_path):
    path = str(tmp_path / "id")
    # what a crashed job leaves: an id file (dead id, old nonces) and request files
    with open(path, "wb") as f:
        f.write(b"\xAA" * 128 + b"\x01\x00\x00\x00\x00\x00\x00\x00" * 64)
    for r in (1, 2):
        with open(path + ".req%d" % r, "wb") as f:
            f.write(b"\x01\x00\x00\x00\x00\x00\x00\x00")
    uid = os.urandom(128)
    out, err = _group(pkg, path, 3, uid, (0.3, 0.0, 0.0))      # the waiters see the stale files first
    assert err == [None] * 3, err
    assert all(o == uid for o in out)                           # never the dead 0xAA id
    # the path is reusable straight away, without any cleanup in between
    uid2 = os.urandom(128)
    out, err = _group(pkg, path, 3, uid2, (0.0, 0.1, 0.0))
    assert err == [None] * 3 and all(o == uid2 for o in out)


def test_rendezvous_times_out_instead_of_hanging(pkg, tmp_path):
    path = str(tmp_path / "id")
    t0 = time.time()
    with pytest.raises(pkg.DQNFatal, match="timed out"):
        pkg.dp_rendezvous_file(path, 1, 2, None, timeout_s=1)   # no rank 0
    with pytest.raises(pkg.DQNFatal, match="timed out"):
        pkg.dp_rendezvous_file(path, 0, 2, os.urandom(128), timeout_s=1)   # no rank 1 (its stale request was cleared)
    assert time.time() - t0 < 10
    with pytest.raises(pkg.DQNFatal, match="bad rank"):
        pkg.dp_rendezvous_file(path, 2, 2, None, timeout_s=1)


def test_product_library_reads_no_environment_variable():
    """A/B switches are dqnhip_config.tuning_flags bits with parity tests, not getenv in the shipped code."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "dqn-hfo_amd", "csrc")
    product = ["learner.hip", "learner_dp.hip", "learner_io.hip", "learner_env.hip", "learner_internal.hip.h", "snapshot.cpp", "dqn_dropin.cpp", "env.hip.h", "gemm_common.hip.h", "gemm_direct.hip.h",
               "hgemm.hip.h", "small_kernels.hip.h"]
    hits = [f for f in product if "getenv" in open(os.path.join(csrc, f)).read()]
    assert hits == [], hits


def test_rccl_build_is_reported_without_a_gpu(pkg):
    """dqnhip_dp_info: which RCCL this process resolved (a PyTorch host process gets torch's bundled build, a bare host
    /opt/rocm's); dqnhip_dp_init cross-checks the version over the group and fails loudly on a mix."""
    ver, path = pkg.DQN.dp_info()
    assert ver > 20000, ver
    assert "rccl" in os.path.basename(path), path
