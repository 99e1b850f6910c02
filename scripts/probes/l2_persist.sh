#!/bin/bash
# Runs scripts/probes/l2_persist_probe.hip (built to scripts/_scratch/l2probe) on the GPU box: plain (event timing), under a kernel trace
# (per-kernel durations of reader<V>), and under one PMC pass each for the L2 hit / miss and fabric-read counters.
#   usage (inside gpurun): scripts/probes/l2_persist.sh  -> gpurun_out/l2_persist.txt
export TMPDIR=/tmp
B=scripts/_scratch/l2probe
O=gpurun_out/l2_persist.txt
mkdir -p gpurun_out
{
  echo "== plain (HIP events around graph replays) =="
  $B 24 300
  echo
  echo "== rocprofv3 --kernel-trace: mean duration per kernel (us) =="
  rm -rf /tmp/l2p; rocprofv3 --kernel-trace --output-format csv -d /tmp/l2p/kt -- $B 24 60 > /dev/null 2>&1
  for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    d=/tmp/l2p/pmc_$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- $B 24 20 > /dev/null 2>&1
  done
  python - <<'PY'
import csv, glob, collections
kt = glob.glob("/tmp/l2p/kt/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(list)
for f in kt:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = v[len(v) // 4:]
    print("  %-40s n %6d  mean %7.3f us  median %7.3f" % (k[:40], len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
print()
print("== PMC, mean per launch ==")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/l2p/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(cnt.items()):
    print("  %-40s %s" % (k[:40], "  ".join("%s %.0f" % (c, sum(v) / len(v)) for c, v in sorted(d.items()))))
PY
} > $O 2>&1
cat $O
