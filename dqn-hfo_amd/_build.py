"""In-tree build of the gfx950 shared library (hipcc, no JIT cache)."""
import fcntl
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdqnhip.so")
_LOCK = os.path.join(CSRC, ".build.lock")


def build(force=False, verbose=False):
    """Bring libdqnhip.so up to date.  `make` itself tracks every source and header of the
    library (SRCS / HDRS in csrc/Makefile), so it is always asked: a stale prebuilt .so can
    never be shipped or tested silently.  Every process that loads the library comes through
    here (N ranks of a torchrun job at once): the make runs under an exclusive file lock, so
    one rank builds while the others wait and then find everything up to date — nobody can
    dlopen a half-written library or race on learner.o."""
    with open(_LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            r = subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), capture_output=True, text=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("hipcc build of libdqnhip.so failed")
    return LIB
