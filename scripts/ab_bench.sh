#!/bin/bash
# A/B/A/B of bench.py under environment toggles, inside ONE gpurun call (same box, same clocks).
#   scripts/ab_bench.sh "<env A>" "<env B>" [extra bench args]
# prints updates/s + ms/update per run
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  for cfg in "$A" "$B"; do
    out=$(env $cfg python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-env "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "[$cfg] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:v[0] for k,v in d["roofline"]["families_us"].items()})')"
  done
done
