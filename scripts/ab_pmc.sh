#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, --kernel-trace only) of the kernels of two
# builds of libdqnhip.so (ab_libs/libA.so, ab_libs/libB.so) under the same bench.py command.
#   usage: scripts/ab_pmc.sh <kernel-name-substring> <bench args...>
export TMPDIR=/tmp
cd /root/repo
pat=$1; shift
L=dqn-hfo_amd/csrc/libdqnhip.so
cp $L /tmp/lib_orig.so
rm -rf /tmp/abpmc; mkdir -p /tmp/abpmc
for v in A B; do
  cp ab_libs/lib$v.so $L
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/abpmc/${v}_$c -- python bench.py "$@" --no-graph --no-cpu-baseline --no-env --no-subrecords --no-live-pmc > /tmp/abpmc/${v}_$c.log 2>&1
    echo "pass $v $c rc=$?"
  done
done
cp /tmp/lib_orig.so $L
python - "$pat" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/abpmc/*/**/*counter_collection.csv", recursive=True):
    v = f.split("/tmp/abpmc/")[1][0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("dqnhip::", "")
        n = (n[:n.index("(")] if "(" in n else n).replace(" ", "")
        if pat in n:
            acc[(v, n)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-4s %-28s %10s %14s %12s %14s" % ("lib", "kernel", "launches", "2xFETCH MB", "WRITE MB", "total MB"))
for (v, n), d in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    print("%-4s %-28s %10d %14.2f %12.2f %14.2f" % (v, n, len(d["FETCH_SIZE"]), 2 * fe / 1024, wr / 1024, (2 * fe + wr) / 1024))
PY
