#!/bin/bash
# Everything profiles/r02_* is made from, in one gpurun call.  Outputs under gpurun_out/r02prof/.
#   kernel stats (rocprofv3 --kernel-trace --stats): fp32 B=256 (headline), fp16 B=4096 (configs[4]), fp16 B=512 (its 8-GPU rank shape), env leg
#   PMC passes (separate runs, --kernel-trace + --pmc only): FETCH_SIZE / WRITE_SIZE, SQ / TCC groups
export TMPDIR=/tmp
O=gpurun_out/r02prof; mkdir -p $O
stats() {  # tag n_updates title -- command...
  tag=$1; nup=$2; title=$3; shift 3
  out=/tmp/prof_$tag; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $O/$tag.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_kernel_stats.csv
  python scripts/stats_to_md.py "$f" $nup "$title" "rocprofv3 --kernel-trace --stats --output-format csv -- $*" > $O/${tag}_kernel_stats.md
}
C="--no-cpu-baseline --no-env --no-subrecords"
stats r02_fp32_b256 0 "rocprofv3 --kernel-trace --stats — round 2, fp32, B=256 (headline; hipGraph replay)" python bench.py --steps 500 --warmup 50 $C
stats r02_fp16_b4096 0 "rocprofv3 --kernel-trace --stats — round 2, fp16 learner, minibatch 4096, 4x1024 (BASELINE configs[4] on one GPU)" python bench.py --precision fp16 --minibatch 4096 --steps 200 --warmup 20 --replay 200000 $C
stats r02_fp16_b512 0 "rocprofv3 --kernel-trace --stats — round 2, fp16 learner, 512 rows (the per-rank shape of configs[4] on 8 GPUs)" python bench.py --precision fp16 --minibatch 512 --steps 500 --warmup 50 --replay 200000 $C
stats r02_env_s68_w64 0 "rocprofv3 --kernel-trace --stats — round 2, batched env front-end, S=68, 64 workers (BASELINE configs[2]); the per-update columns are per batched env step (1000 steps in the trace)" python scripts/env_probe.py 68 64
# PMC
scripts/pmc_fetch.sh > $O/pmc_fetch.log 2>&1; cp gpurun_out/pmc_fetch.json $O/ 2>/dev/null
scripts/pmc_summary.sh > $O/pmc_summary.log 2>&1; cp gpurun_out/pmc_summary.json $O/ 2>/dev/null
ls -la $O
