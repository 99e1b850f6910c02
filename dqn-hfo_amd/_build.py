"""In-tree build of the gfx950 shared library (hipcc, no JIT cache)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdqnhip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", "Makefile"))]
    deps.append(os.path.join(CSRC, "..", "..", "include", "dqnhip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile libdqnhip.so for gfx950 if missing or older than its sources."""
    if force or _stale():
        r = subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), capture_output=True, text=True)
        if verbose or r.returncode:
            print(r.stdout[-4000:], r.stderr[-4000:])
        if r.returncode:
            raise RuntimeError("hipcc build of libdqnhip.so failed")
    return LIB
