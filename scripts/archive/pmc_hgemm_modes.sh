#!/bin/bash
# PMC passes (separate runs per counter group, --kernel-trace + --pmc only) on the fp16 GEMM in its three uses at 4096 rows:
# forward (k-major operands, 128x128), dgrad-like/wgrad-like with reduction-major operands (transposing LDS reads).
export TMPDIR=/tmp
echo "# rocprofv3 PMC, fp16 GEMM kernels (scripts/pmc_hgemm_modes.sh; per-launch means)"
for args in "4 1 4096 1024 1024 30:forward 4096x1024x1024, k-major operands, 128x128 tile" "2 2 1024 1024 4096 30:wgrad-shaped 1024x1024 K=4096, k-major operands (round-1 form), 64x64 split-K tile" "6 2 1024 1024 4096 30:wgrad-shaped 1024x1024 K=4096, reduction-major operands (ds_read_b64_tr_b16), 64x64 split-K tile"; do
  a="${args%%:*}"; t="${args#*:}"
  echo; echo "## $t  (hgemm_one.py $a)"
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    out=/tmp/pmc_$RANDOM
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python scripts/hgemm_one.py $a > /tmp/pmc.log 2>&1
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "hgemm_nt" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("- %-32s %.4g" % (k, sum(v) / len(v)))
PY
  done
done
