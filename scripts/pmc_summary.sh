#!/bin/bash
# Rebuilds profiles-style PMC summary (per-kernel means per launch) -> gpurun_out/pmc_summary.json
# (FETCH_SIZE / WRITE_SIZE: scripts/pmc_fetch.sh, one counter per pass — both in one pass hang rocprofv3 here)
# Separate rocprofv3 passes per counter group, --kernel-trace only (no other trace domains).
export TMPDIR=/tmp
rm -rf /tmp/pmcs; mkdir -p /tmp/pmcs
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"; timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs/p$i -- python bench.py --steps 60 --warmup 10 --no-graph --no-cpu-baseline --no-env --no-subrecords --replay 200000 > /tmp/pmcs/log$i.txt 2>&1; echo "  rc=$?"
done
python - <<'PY'
import csv, glob, json, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcs/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dqnhip::" not in n:
            continue
        n = n.replace("void ", "").replace("dqnhip::", "")
        n = n[:n.index("(")] if "(" in n else n
        n = n.replace(" ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python bench.py --steps 60 --warmup 10 --no-graph --no-cpu-baseline --no-env --no-subrecords --replay 200000 (MI355X, round 1, scripts/pmc_summary.sh)",
       "units": {"FETCH_SIZE": "KB as reported (double it for wide coalesced reads on gfx950, MI355X_MICROARCH.md §HBM)", "WRITE_SIZE": "KB",
                 "SQ_VALU_MFMA_BUSY_CYCLES": "cycles summed over SIMDs (32 per v_mfma_f32_16x16x4_f32)",
                 "SQ_WAVE_CYCLES/SQ_WAIT_*": "quad-cycles summed over waves"},
       "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in sorted(acc.items())}}
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
for k, d in out["kernels"].items():
    if "FETCH_SIZE" in d:
        print("%-34s fetch %.0f KB (x2 = %.2f MB)  write %.0f KB" % (k, d["FETCH_SIZE"], 2 * d["FETCH_SIZE"] / 1024, d["WRITE_SIZE"]))
PY
