#!/bin/bash
# A/B/A/B... of option sets on one box: scripts/ab.sh N d reps rounds "optsA" "optsB" ["optsC" ...]
cd $GRAFT_REPO_ROOT
N=$1; D=$2; REPS=$3; ROUNDS=$4; shift 4
for r in $(seq 1 $ROUNDS); do
  for o in "$@"; do
    timeout -k 10 180 python scripts/prof_target.py $N $D 1024 $REPS timing=0 $o > gpurun_out/ab.tmp 2>/dev/null || exit 1
    echo "[$o] $(sed -n 1p gpurun_out/ab.tmp | sed 's/nlml.*//')"
  done
done
