#!/bin/bash
# A/B/A/B of bench.py under two values of dqnhip_config.tuning_flags, inside ONE gpurun call (same box, same clocks).
#   scripts/ab_bench.sh <flags A> <flags B> [extra bench args]      e.g.  scripts/ab_bench.sh 0 1 --precision fp16 --minibatch 4096
# prints updates/s + ms/update per run.  (The library reads no environment variable: include/dqnhip.h DQNHIP_TUNE_*.)
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  for cfg in "$A" "$B"; do
    out=$(python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-env --no-subrecords --tuning $cfg "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "[tuning=$cfg] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:v[0] for k,v in d["roofline"]["families_us"].items()})')"
  done
done
