#!/bin/bash
# One update's kernel timeline (start, duration, gap to the previous kernel's end) from a rocprofv3 kernel trace.
#   usage (inside gpurun): scripts/probes/timeline.sh <tag> <bench args...>  -> gpurun_out/<tag>_timeline.txt
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/tl_$tag
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python bench.py "$@" > /tmp/bench_tl_$tag.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
mkdir -p gpurun_out
python - "$f" > gpurun_out/${tag}_timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# updates are delimited by k_gather launches; take the median-length one among the last 200
gi = [i for i, n in enumerate(names) if "k_gather" in n]
spans = []
for a, b in zip(gi[-202:-1], gi[-201:]):
    spans.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]), a, b))
spans.sort()
wall, a, b = spans[len(spans) // 2]
print("median update: %.2f us wall, %d launches" % (wall / 1e3, b - a))
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None; sd = sg = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    nm = r["Kernel_Name"].replace("dqnhip::", "").replace("void ", "")
    nm = nm[:nm.index("(")] if "(" in nm else nm
    print("%8.2f  dur %6.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, nm))
    if r is not rows[b]: sd += (e - s) / 1e3; sg += gap
    prev_end = e
print("sum of durations %.2f us, sum of gaps %.2f us" % (sd, sg))
# average gap after each kernel type over the last 200 updates
acc = collections.defaultdict(list)
for i in range(gi[-202], gi[-1]):
    nm = names[i].replace("dqnhip::", "").replace("void ", ""); nm = nm[:nm.index("(")] if "(" in nm else nm
    acc[nm].append((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3)
print("\naverage gap AFTER a kernel (us), last 200 updates:")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("  %-40s n/update %.1f  gap %.2f" % (k, len(v) / 201.0, sum(v) / len(v)))
PY
cat gpurun_out/${tag}_timeline.txt
