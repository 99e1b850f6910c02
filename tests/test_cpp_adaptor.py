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
