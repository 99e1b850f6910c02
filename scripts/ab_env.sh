#!/bin/bash
# same-box A/B of two builds (ab_libs/libA.so, libB.so) on the batched env front-end.   usage: scripts/ab_env.sh [reps] [S] [workers]
cd /root/repo
reps=${1:-3}; S=${2:-68}; W=${3:-64}
L=dqn-hfo_amd/csrc/libdqnhip.so; cp $L /tmp/lib_orig.so
for rep in $(seq $reps); do for v in A B; do cp ab_libs/lib$v.so $L; echo "[$v] $(python scripts/env_probe.py $S $W 2>&1 | tail -1)"; done; done
cp /tmp/lib_orig.so $L
