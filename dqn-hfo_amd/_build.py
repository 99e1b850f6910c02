"""In-tree build of the gfx950 shared library (hipcc, no JIT cache)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdqnhip.so")
TEST_LIB = os.path.join(CSRC, "libdqnhip_test.so")


def build(force=False, verbose=False):
    """Bring libdqnhip.so up to date.  `make` itself tracks every source and header of the
    library (SRCS / HDRS in csrc/Makefile), so it is always asked: a stale prebuilt .so can
    never be shipped or tested silently."""
    r = subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("hipcc build of libdqnhip.so failed")
    return LIB
