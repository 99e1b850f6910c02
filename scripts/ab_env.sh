#!/bin/bash
# Same-box A/B of two builds of libdqnhip.so (ab_libs/libA.so, libB.so) under scripts/env_probe.py.   usage: scripts/ab_env.sh [reps] [S workers...]
cd /root/repo
reps=${1:-3}; shift
L=dqn-hfo_amd/csrc/libdqnhip.so
cp $L /tmp/lib_orig.so
for rep in $(seq $reps); do
  for v in A B; do
    cp ab_libs/lib$v.so $L
    echo "[$v] $(python scripts/env_probe.py "$@" 2>/dev/null | tr '\n' ';')"
  done
done
cp /tmp/lib_orig.so $L
