"""The C-ABI library loads (no GPU needed: hipcc cross-compiles, dlopen works without a device)
and exports every symbol that include/*.h declares; the ctypes table in capi.py covers the same
set.  No compute calls."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import testlib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


INTERNAL = os.path.join(ROOT, "tests", "csrc", "dqnhip_internal.h")      # test/tuning hooks: exported by tests/csrc/libdqnhip_test.so, never by the product library


def declared_functions(internal=False):
    names = set()
    for hdr in ([INTERNAL] if internal else glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = open(hdr).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"\b(dqnhip_[a-z_0-9]+)\s*\(", src):
            names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported(pkg):
    lib_path = pkg.build()
    decl = declared_functions()
    assert len(decl) >= 30
    lib = C.CDLL(lib_path)
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(dqnhip_[a-z_0-9]+)\b", out))
    assert decl <= exported
    # the test hooks live in their own library and are NOT in the product library
    tdecl = declared_functions(internal=True)
    assert tdecl == {"dqnhip_test_gemm", "dqnhip_test_hgemm", "dqnhip_test_adam", "dqnhip_test_chain",
                     "dqnhip_test_loadpath", "dqnhip_test_hgemm_backward", "dqnhip_test_overlap", "dqnhip_test_launch_floor"}
    assert not (tdecl & exported)
    tlib = testlib.load_test()
    assert all(hasattr(tlib, n) for n in tdecl)
    # the product library links RCCL itself (native data parallelism, dqnhip_dp_*)
    needed = subprocess.run(["readelf", "-d", lib_path], capture_output=True, text=True).stdout
    assert "librccl" in needed


def test_ctypes_table_matches_header(pkg):
    decl = declared_functions()
    assert set(pkg.capi.SIGNATURES) == decl


def test_config_struct_size_and_defaults(pkg):
    """dqnhip_default_config is pure host code: the defaults are the reference's
    (src/dqn.hpp:19, src/dqn.cpp:21-31, 425, src/dqn_main.cpp:30-37)."""
    lib = pkg.capi.load()
    cfg = pkg.capi.Config()
    lib.dqnhip_default_config(C.byref(cfg), 59)
    assert cfg.struct_size == C.sizeof(pkg.capi.Config)
    assert (cfg.minibatch, cfg.state_size, cfg.num_hidden) == (32, 59, 4)
    assert list(cfg.hidden)[:4] == [1024, 512, 256, 128]
    assert cfg.replay_capacity == 500000 and cfg.soft_update_freq == 1
    assert (cfg.gamma, cfg.beta, cfg.tau) == (0.99, 0.5, 0.001)
    assert abs(cfg.actor_lr - 1e-5) < 1e-12 and abs(cfg.critic_lr - 1e-3) < 1e-10
    assert abs(cfg.momentum - 0.95) < 1e-7 and abs(cfg.momentum2 - 0.999) < 1e-7
    assert abs(cfg.delta - 1e-8) < 1e-15 and cfg.clip_gradients == 10.0
    assert lib.dqnhip_grad_arena_bytes(C.byref(cfg)) > 4 * (751754 + 760833)
    cfg.minibatch = 33                                   # not a multiple of 32 -> rejected
    assert lib.dqnhip_grad_arena_bytes(C.byref(cfg)) == 0
    assert b"multiple of 32" in lib.dqnhip_last_error()


def test_create_fails_loudly_without_gpu(pkg):
    """On a box with no HIP device the product path must error, never fall back."""
    import pytest
    from conftest import _gpu_present
    if _gpu_present():
        pytest.skip("GPU present")
    with pytest.raises(pkg.DQNFatal):
        pkg.DQN(59)
