"""The C++ `dqn::DQN` adaptor (dqn-hfo_amd/csrc/dqn_adaptor.hpp) compiles against the C-ABI with
plain g++ (CPU) and drives the learner like src/dqn_main.cpp does (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adaptor_smoke")


def _build(pkg):
    lib = pkg.build()
    src = os.path.join(ROOT, "tests", "cpp", "adaptor_smoke.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, src, lib, "-Wl,-rpath," + os.path.dirname(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_adaptor_compiles_and_links(pkg):
    _build(pkg)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adaptor_runs_episode_loop(pkg, gpu):
    exe = _build(pkg)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "adaptor smoke OK" in r.stdout
