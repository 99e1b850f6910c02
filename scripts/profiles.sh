#!/bin/bash
# Everything profiles/rNN_* is made from, in one gpurun call.  Outputs under gpurun_out/rNNprof/; fails (exit 1) if a
# table it was asked for comes out empty, so that DESIGN never cites a 0-byte file again.
#   usage: ROUND=r04 scripts/profiles.sh [stats|pmc|all]        (then copy gpurun_out/$ROUND"prof"/$ROUND_* into profiles/)
export TMPDIR=/tmp
what=${1:-all}
R=${ROUND:-r05}; N=${R#r}; N=$((10#$N))
O=gpurun_out/${R}prof; mkdir -p $O
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
rc=0
stats() {  # tag n_updates title -- command...
  tag=$1; nup=$2; title=$3; shift 3
  out=/tmp/prof_$tag; rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $O/$tag.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then echo "NO kernel_stats.csv for $tag"; rc=1; return; fi
  cp "$f" $O/${tag}_kernel_stats.csv
  python scripts/stats_to_md.py "$f" $nup "$title" "rocprofv3 --kernel-trace --stats --output-format csv -- $*" > $O/${tag}_kernel_stats.md
  if [ $(grep -c '^| `' $O/${tag}_kernel_stats.md) -lt 3 ]; then echo "EMPTY table for $tag"; rc=1; fi
}
C="--no-cpu-baseline --no-env --no-subrecords --no-live-pmc --no-live-trace"
if [ "$what" != pmc ]; then
stats ${R}_fp32_b256 0 "rocprofv3 --kernel-trace --stats — round $N, fp32, B=256 (headline; hipGraph replay)" python bench.py --steps 500 --warmup 50 $C
stats ${R}_fp16_b4096 0 "rocprofv3 --kernel-trace --stats — round $N, fp16 learner, minibatch 4096, 4x1024 (BASELINE configs[4] on one GPU)" python bench.py --precision fp16 --minibatch 4096 --steps 200 --warmup 20 --replay 200000 $C
stats ${R}_fp16_b512 0 "rocprofv3 --kernel-trace --stats — round $N, fp16 learner, 512 rows (the per-rank shape of configs[4] on 8 GPUs)" python bench.py --precision fp16 --minibatch 512 --steps 500 --warmup 50 --replay 200000 $C
stats ${R}_env_s68_w64 0 "rocprofv3 --kernel-trace --stats — round $N, batched env front-end, S=68, 64 workers (BASELINE configs[2]); the per-update columns are per batched env step (1000 steps in the trace)" python scripts/env_probe.py 68 64
stats ${R}_env_s58_w2048 0 "rocprofv3 --kernel-trace --stats — round $N, batched env front-end, S=58, 2048 workers (BASELINE configs[4]); per batched env step" python scripts/env_probe.py 58 2048
fi
if [ "$what" != stats ]; then
ROUND=$R scripts/pmc.sh > $O/pmc.log 2>&1
[ -s $O/${R}_pmc_summary.json ] || { echo "EMPTY ${R}_pmc_summary.json"; rc=1; }
python -c "import json,sys; j=json.load(open('$O/${R}_pmc_summary.json')); sys.exit(0 if len(j['kernels'])>5 and len(j['kernels_fp16_b4096'])>5 else 1)" || { echo "PMC summary has too few kernels"; rc=1; }
fi
ls -la $O
exit $rc
