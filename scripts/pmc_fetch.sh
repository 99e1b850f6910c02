#!/bin/bash
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  echo "pass $c"; rm -rf /tmp/pf_$c
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pf_$c -- python bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline --no-env --no-subrecords --replay 100000 > /tmp/pf_$c.log 2>&1; echo "  rc=$?"
done
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dqnhip::" not in n: continue
        n = n.replace("void ", "").replace("dqnhip::", ""); n = n[:n.index("(")] if "(" in n else n; n = n.replace(" ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("gpurun_out/pmc_fetch.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k, d)
PY
