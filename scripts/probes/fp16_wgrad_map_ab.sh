#!/bin/bash
# fp16 learner at minibatch 4096 (configs[4] on one GPU): the grouped wgrad launch with the XCD-aware map (default) against the
# flat map (--tuning 8 = DQNHIP_TUNE_FP16_WGRAD_FLAT_MAP): update time (graph replay, alternating), per-kernel durations
# (rocprofv3 --kernel-trace) and fabric traffic (one --pmc pass each for FETCH_SIZE / WRITE_SIZE, eager).
#   usage (inside gpurun): scripts/probes/fp16_wgrad_map_ab.sh -> gpurun_out/fp16_wgrad_map_ab.txt
export TMPDIR=/tmp
O=gpurun_out/fp16_wgrad_map_ab.txt; mkdir -p gpurun_out
C="--precision fp16 --minibatch 4096 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc --no-live-trace --replay 200000"
{
  echo "== update time, ms per update (bench.py, graph replay; three alternations) =="
  for rep in 1 2 3; do
    for t in 0 8; do
      python bench.py $C --steps 200 --warmup 40 --tuning $t 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.readlines()[-1]); print('  tuning %s  %.5f ms' % ('$t', j['ms_per_step']))"
    done
  done
  for t in 0 8; do
    rm -rf /tmp/fab_$t
    rocprofv3 --kernel-trace --output-format csv -d /tmp/fab_$t/kt -- python bench.py $C --steps 64 --warmup 32 --tuning $t --trace-child > /dev/null 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/fab_$t/$c -- python bench.py $C --steps 10 --warmup 3 --no-graph --prewarm-ms 0 --tuning $t --trace-child > /dev/null 2>&1
    done
  done
  python - <<'PY'
import csv, glob, collections
for t in (0, 8):
    print("\n== tuning %d: %s ==" % (t, "XCD-aware map (default)" if t == 0 else "flat map (round 4)"))
    dur = collections.defaultdict(list)
    for f in glob.glob("/tmp/fab_%d/kt/**/*kernel_trace.csv" % t, recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows = rows[len(rows) // 2:]
        for r in rows:
            n = r["Kernel_Name"].replace("void ", "").replace("dqnhip::", ""); n = n[:n.index("(")] if "(" in n else n
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("/tmp/fab_%d/%s/**/*counter_collection.csv" % (t, c), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("void ", "").replace("dqnhip::", ""); n = n[:n.index("(")] if "(" in n else n
                cnt[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in sorted(dur, key=lambda k: -sum(dur[k])):
        if "hgemm" not in n: continue
        fs = cnt[n].get("FETCH_SIZE", []); ws = cnt[n].get("WRITE_SIZE", [])
        tr = (2 * sum(fs) / len(fs) + (sum(ws) / len(ws) if ws else 0)) / 1024 if fs else float("nan")
        print("  %-34s n %4d  mean %7.2f us   traffic 2 x FETCH + WRITE = %7.1f MB per launch" % (n[:34], len(dur[n]), sum(dur[n]) / len(dur[n]), tr))
PY
} > $O 2>&1
cat $O
