#!/bin/bash
# Same-box A/B of two dqnhip_config.tuning_flags values under bench.py inside ONE gpurun call.
#   usage: scripts/ab_tuning.sh <flagsA> <flagsB> [reps] [extra bench args]
cd /root/repo
A=${1:-0}; B=${2:-2}; reps=${3:-3}; shift 3
for rep in $(seq $reps); do
  for v in $A $B; do
    out=$(python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc --tuning $v "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "[tuning $v] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); f=d["roofline"]["families_us"]; print(d["value"], d["ms_per_step"], {k: v[0] for k, v in f.items()})')"
  done
done
