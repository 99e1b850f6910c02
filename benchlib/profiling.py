"""Live rocprofv3 child passes of bench.py (kernel trace of the graph-replayed update; FETCH_SIZE / WRITE_SIZE) and the committed
PMC summary they fall back to."""
import json
import os
import sys

from .flops import KERNEL_NAMES  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


PMC_SUMMARY = "profiles/r06_pmc_summary.json"


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel`.  PMC counters cannot be read from inside this process
    (rocprofv3 wraps the command), so the figure comes from the committed PMC passes of the SAME
    kernels (scripts/pmc.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    bench.py; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on
    gfx950) and is labelled with its source.  (None, None) if that kernel is not in the file."""
    try:
        j = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
        d = j["kernels"][kernel.replace(" ", "")]
        return int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024), "%s (%s)" % (PMC_SUMMARY, j.get("kernels_version", "kernel version not recorded"))
    except Exception:
        return None, None


def live_pmc_traffic(kernel, extra_args):
    """HBM-side bytes per launch of `kernel`, measured NOW on this box: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc <one counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, no other trace domain, as
    MI355X_MICROARCH.md prescribes), eager launches, a small replay.  2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950
    correction for wide coalesced reads).  Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    want = kernel.replace(" ", "")
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dqnhip_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               BENCH, "--steps", "20", "--warmup", "5", "--prewarm-ms", "0", "--no-graph", "--no-cpu-baseline",
               "--no-env", "--no-subrecords", "--no-live-pmc", "--replay", "100000"] + list(extra_args)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        except Exception as e:          # noqa: BLE001
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 pass %s failed: %r" % (counter, e)
        acc = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                n = row["Kernel_Name"].replace("void ", "").replace("dqnhip::", "")
                n = (n[:n.index("(")] if "(" in n else n).replace(" ", "")
                if n == want and row["Counter_Name"] == counter:
                    acc.append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not acc:
            return None, "rocprofv3 pass %s: no rows for %s (rc %d)" % (counter, want, r.returncode)
        vals[counter] = sum(acc) / len(acc)
    return int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), (
        "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of this bench.py, eager, "
        "%d + %d launches averaged); 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md" % (len(acc), len(acc)))


def live_kernel_trace(extra_args, updates=320):
    """Per-kernel launch durations of the GRAPH-REPLAYED update, measured NOW on this box: one child run of this script under
    `rocprofv3 --kernel-trace` (no counters, no other trace domain), the K steps enqueued exactly as the headline enqueues them
    (dqnhip_update_async_n: sixteen updates per hipGraph launch).  Inside a replayed graph a kernel's reported duration runs up
    to the next kernel's start (profiles/r04_graph_gap.txt), so these durations are what the wall clock is made of — the same
    numbers `rocprofv3 --kernel-trace --stats` prints (profiles/rNN_fp32_b256_kernel_stats.md); eager HIP events around single
    launches (the fallback) differ from them by a few per cent and from run to run.
    Returns ({kernel: (mean_us, launches_per_update)}, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="dqnhip_kt_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, BENCH,
           "--steps", str(updates), "--warmup", "32", "--no-cpu-baseline", "--no-env", "--no-subrecords", "--no-live-pmc", "--no-live-trace", "--trace-child",
           "--replay", "100000"] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    except Exception as e:              # noqa: BLE001
        shutil.rmtree(d, ignore_errors=True)
        return None, "rocprofv3 --kernel-trace child failed: %r" % (e,)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"].replace("void ", "").replace("dqnhip::", "")
            n = (n[:n.index("(")] if "(" in n else n).replace(" ", "")
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), n))
    shutil.rmtree(d, ignore_errors=True)
    if not rows:
        return None, "rocprofv3 --kernel-trace child: no kernel rows (rc %d: %s)" % (r.returncode, (r.stderr or "")[-200:])
    rows.sort()
    # the timed region = the last `updates` updates: count them by the actor's optimiser pass, the last launch of every update
    # (k_adam_soft_gather inside the sixteen-update graph, k_adam_soft with the tick otherwise: two k_adam_soft* per update)
    adam = [i for i, x in enumerate(rows) if x[2].startswith("k_adam_soft")]
    if len(adam) < 2 * updates:
        return None, "rocprofv3 --kernel-trace child: %d optimiser launches < 2 x %d updates" % (len(adam), updates)
    first = adam[len(adam) - 2 * updates - 1] + 1 if len(adam) > 2 * updates else 0
    acc = {}
    for s0, e0, n in rows[first:]:
        a = acc.setdefault(n, [0.0, 0])
        a[0] += (e0 - s0) / 1e3; a[1] += 1
    span_us = (rows[-1][1] - rows[first][0]) / 1e3
    out = {n: (v[0] / v[1], v[1] / float(updates)) for n, v in acc.items()}
    return out, ("measured in this run: rocprofv3 --kernel-trace of a child pass of this bench.py (graph replay, %d updates, %.2f us per update "
                 "inside the trace); a kernel's duration inside a replayed graph runs up to the next kernel's start" % (updates, span_us / updates))
