#!/bin/bash
# PMC passes on one hgemm shape (separate runs per counter group, --kernel-trace only)
export TMPDIR=/tmp
args="4 1 4096 1024 4096 30"
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  out=/tmp/pmc_$RANDOM
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python scripts/hgemm_one.py $args > /tmp/pmc.log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "hgemm_nt" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-32s per-launch mean %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
