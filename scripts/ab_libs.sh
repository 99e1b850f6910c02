#!/bin/bash
# Same-box A/B of two builds of libdqnhip.so (ab_libs/libA.so, ab_libs/libB.so: copy them there after building each variant):
# alternates them under bench.py inside ONE gpurun call.   usage: scripts/ab_libs.sh [reps] [extra bench args]
cd /root/repo
reps=${1:-3}; shift
L=dqn-hfo_amd/csrc/libdqnhip.so
cp $L /tmp/lib_orig.so
for rep in $(seq $reps); do
  for v in A B; do
    cp ab_libs/lib$v.so $L
    out=$(python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "[$v] $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); f=d["roofline"]["families_us"]; print(d["value"], d["ms_per_step"], {k: v[0] for k, v in f.items()})')"
  done
done
cp /tmp/lib_orig.so $L
