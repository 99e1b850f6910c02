#!/bin/bash
# usage: scripts/prof_stats.sh <tag> <bench args...>   -> gpurun_out/<tag>_kernel_stats.csv + printed top kernels
tag=$1; shift
export TMPDIR=/tmp
out=/tmp/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py "$@" > /tmp/bench_$tag.log 2>&1
tail -2 /tmp/bench_$tag.log | cut -c1-600
f=$(find $out -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out && cp "$f" gpurun_out/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print("%-72s calls %6s avg_us %9.2f pct %6s" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
