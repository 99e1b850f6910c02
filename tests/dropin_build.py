"""Builds the reference's UNCHANGED driver against this repo's drop-in: src/dqn_main.cpp and
src/hfo_game.cpp are compiled where they lie under /root/reference (through symlinks in a scratch
directory, so that `#include "dqn.hpp"` resolves to include/dqn.hpp and not to the reference's
header next to the driver — the same effect as deleting src/dqn.{hpp,cpp} in the reference tree,
INTEGRATION.md), together with dqn-hfo_amd/csrc/dqn_dropin.cpp, against include/shim/ (gflags,
glog, boost, caffe, HFO are absent from this image) and libdqnhip.so.

Output: tests/cpp/_dropin/dqn (git-ignored; it travels to the GPU box with the snapshot, where
/root/reference does not exist).  This is a source-compatibility / end-to-end demonstration of
the boundary, NOT an oracle: nothing compares numbers against it.
"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_dropin")
EXE = os.path.join(OUT_DIR, "dqn")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "shim")]


def reference_present():
    return os.path.isfile(os.path.join(REF, "dqn_main.cpp"))


def syntax_only(extra_args, source):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + extra_args + [source], capture_output=True, text=True)


def build(lib):
    """-> path of the driver binary (raises on failure)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("dqn_main.cpp", "hfo_game.cpp"):
            os.symlink(os.path.join(REF, f), os.path.join(tmp, f))
        objs = []
        for src, inc in ((os.path.join(tmp, "dqn_main.cpp"), INC + ["-I" + REF]),
                         (os.path.join(tmp, "hfo_game.cpp"), INC + ["-I" + REF]),
                         (os.path.join(ROOT, "dqn-hfo_amd", "csrc", "dqn_dropin.cpp"), INC + ["-I" + REF])):
            obj = os.path.join(tmp, os.path.basename(src) + ".o")
            r = subprocess.run(["g++", "-std=c++17", "-O1", "-c", "-o", obj, src] + inc, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError("compile %s:\n%s" % (src, r.stderr[-4000:]))
            objs.append(obj)
        r = subprocess.run(["g++", "-o", EXE] + objs + [lib, "-Wl,-rpath," + os.path.dirname(lib), "-lpthread"],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link:\n%s" % r.stderr[-4000:])
    return EXE
