"""Build + dlopen of the test / tuning harness tests/csrc/libdqnhip_test.so (tests/csrc/dqnhip_internal.h): kernel-level
tests and probes.  Tests and scripts only — nothing under dqn-hfo_amd/ knows it exists."""
import ctypes as C
import fcntl
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
TEST_LIB = os.path.join(CSRC, "libdqnhip_test.so")
_lib = None


def build_test(verbose=False):
    """`make` under an exclusive file lock (several test processes may arrive at once), always consulted: a stale
    prebuilt library is never tested silently."""
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            r = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("hipcc build of libdqnhip_test.so failed")
    return TEST_LIB


def load_test():
    global _lib
    if _lib is None:
        build_test()
        _lib = C.CDLL(TEST_LIB)
    return _lib
