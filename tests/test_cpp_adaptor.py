"""include/dqn.hpp + dqn-hfo_amd/csrc/dqn_dropin.cpp (the `dqn::DQN` surface of src/dqn.hpp over the C-ABI)
compile with plain g++ against include/shim/ (CPU) and drive the learner like src/dqn_main.cpp does (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adaptor_smoke")


def _build(pkg):
    lib = pkg.build()
    src = os.path.join(ROOT, "tests", "cpp", "adaptor_smoke.cpp")
    dropin = os.path.join(ROOT, "dqn-hfo_amd", "csrc", "dqn_dropin.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "shim"),
           "-o", EXE, src, dropin, lib, "-Wl,-rpath," + os.path.dirname(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_adaptor_compiles_and_links(pkg):
    _build(pkg)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adaptor_runs_episode_loop(pkg, gpu):
    exe = _build(pkg)
    r = subprocess.run([exe, "-seed", "7", "-memory", "5000", "-memory_threshold", "100"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "adaptor smoke OK" in r.stdout


@pytest.mark.gpu
def test_chained_updates_change_nothing_the_driver_sees(pkg, gpu):
    """-chained_updates (default on): UpdateActorCritic() predicts the next call's indices from a copy of its std::mt19937 and lets the
    library run the next gather / first layers ahead.  The engine's call order, every loss and every weight must be those of the
    plain blocking form: the episode loop (epsilon draws between the bursts break every prediction) ends on the same numbers."""
    exe = _build(pkg)
    outs = []
    for flag in ("-chained_updates=true", "-chained_updates=false"):
        r = subprocess.run([exe, "-seed", "7", "-memory", "5000", "-memory_threshold", "100", flag], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (flag, r.returncode, r.stdout, r.stderr)
        outs.append([l for l in r.stdout.splitlines() if "adaptor smoke OK" in l])
    assert outs[0] == outs[1] and len(outs[0]) == 1, outs


def test_cpu_mode_stops_with_the_adaptors_message(pkg):
    """-gpu=false (Caffe CPU mode, src/dqn_main.cpp:208-212): no CPU backend exists by design; the adaptor says so through the
    driver's logging path BEFORE touching the device (so this runs on a box without a GPU), instead of a bare CHECK."""
    exe = _build(pkg)
    r = subprocess.run([exe, "-check", "cpu_mode"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.returncode != 20, (r.returncode, r.stdout, r.stderr)
    assert "-gpu=false" in r.stderr and "not provided by the MI355X drop-in" in r.stderr and "src/dqn_main.cpp:208-212" in r.stderr, r.stderr


@pytest.mark.gpu
def test_select_actions_keeps_the_references_batch_cap(pkg, gpu):
    """src/dqn.cpp:699 CHECK_LE(states_batch.size(), kMinibatchSize): kept (with this learner's -minibatch); -select_actions_cap widens it."""
    exe = _build(pkg)
    r = subprocess.run([exe, "-check", "select_cap", "-memory", "5000"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "select_cap:" not in r.stdout and "Check failed" in r.stderr, (r.returncode, r.stdout, r.stderr)
    for cap in ("-1", "64"):
        r = subprocess.run([exe, "-check", "select_cap", "-memory", "5000", "-select_actions_cap", cap], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "select_cap: 33 actions" in r.stdout, (cap, r.returncode, r.stdout, r.stderr)
