#!/bin/bash
# PMC passes over the fp32 update (separate runs per counter group, --kernel-trace only); per-kernel means
export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  out=/tmp/pmcb_$RANDOM
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python bench.py --no-cpu-baseline --no-env --no-subrecords --steps 60 --warmup 10 --replay 100000 --no-graph > /tmp/pmcb.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_" in n or "adam" in n:
        n = n.replace("void dqnhip::", "").split("(")[0]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(acc):
    print(n, {k: "%.4g" % (sum(v) / len(v)) for k, v in acc[n].items()})
PY
done
