#!/bin/bash
# PMC passes -> gpurun_out/rNNprof/rNN_pmc_summary.json (per-kernel means per launch); ROUND=r04 (default) names the round.
# One rocprofv3 run per counter group, --kernel-trace + --pmc only (no other trace domain: gpurun refuses the mix),
# FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   workloads: the fp32 headline (B = 256) and the fp16 learner at minibatch 4096 (configs[4] on one GPU), eager launches
export TMPDIR=/tmp
R=${ROUND:-r05}
export R
O=gpurun_out/${R}prof; mkdir -p $O
rm -rf /tmp/pmc3; mkdir -p /tmp/pmc3
C="--no-graph --no-cpu-baseline --no-env --no-subrecords --no-live-pmc --no-live-trace --replay 200000"
i=0
for wl in "fp32_b256:--steps 60 --warmup 10" "fp16_b4096:--precision fp16 --minibatch 4096 --steps 30 --warmup 5"; do
  tag=${wl%%:*}; args=${wl#*:}
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    echo "pass $i [$tag]: $grp"
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc3/${tag}_p$i -- python bench.py $args $C > /tmp/pmc3/log$i.txt 2>&1; echo "  rc=$?"
  done
done
python - <<'PY'
import csv, glob, json, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob("/tmp/pmc3/*/**/*counter_collection.csv", recursive=True):
    wl = "fp16_b4096" if "/fp16_b4096_p" in f else "fp32_b256"
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dqnhip::" not in n and "k_env" not in n:
            continue
        n = n.replace("void ", "").replace("dqnhip::", "")
        n = n[:n.index("(")] if "(" in n else n
        n = n.replace(" ", "")
        acc[wl][n][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(d): return {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(d.items())}
head = open(".git_head").read().strip() if os.path.exists(".git_head") else "unknown"
kv = json.load(open(".kernel_versions.json")) if os.path.exists(".kernel_versions.json") else {}
R = os.environ.get("R", "r04")
out = {"source": "scripts/pmc.sh: rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python bench.py ... --no-graph (MI355X, round %s)" % R[1:].lstrip("0"),
       "head": head,
       "kernels_version": "HEAD %s; kernel sources last changed at: %s" % (head, ", ".join("%s %s" % (k, v) for k, v in sorted(kv.items())) or "not recorded"),
       "units": {"FETCH_SIZE": "KB as reported (double it for wide coalesced reads on gfx950, MI355X_MICROARCH.md HBM section)", "WRITE_SIZE": "KB",
                 "SQ_VALU_MFMA_BUSY_CYCLES": "cycles summed over SIMDs", "SQ_WAVE_CYCLES/SQ_WAIT_*": "quad-cycles summed over waves"},
       "kernels": mean(acc["fp32_b256"]), "kernels_fp16_b4096": mean(acc["fp16_b4096"])}
json.dump(out, open("gpurun_out/%sprof/%s_pmc_summary.json" % (R, R), "w"), indent=1)
for wl in ("kernels", "kernels_fp16_b4096"):
    print(wl)
    for k, d in out[wl].items():
        if "FETCH_SIZE" in d:
            print("  %-34s fetch %.0f KB (x2 = %.2f MB)  write %.0f KB" % (k, d["FETCH_SIZE"], 2 * d["FETCH_SIZE"] / 1024, d.get("WRITE_SIZE", float("nan"))))
PY
