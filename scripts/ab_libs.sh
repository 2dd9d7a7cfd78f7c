#!/bin/bash
# same-box A/B of two BUILDS of libgpk.so: scripts/ab_libs.sh N d reps rounds libA libB [opts]   (the chosen file is copied over gptest_amd/libgpk.so per run)
cd $GRAFT_REPO_ROOT
N=$1; D=$2; REPS=$3; ROUNDS=$4; A=$5; B=$6; shift 6
for r in $(seq 1 $ROUNDS); do
  for lib in $A $B; do
    cp $lib gptest_amd/libgpk.so
    timeout -k 10 180 python scripts/prof_target.py $N $D 1024 $REPS timing=0 "$@" > gpurun_out/ab.tmp 2>/dev/null || exit 1
    echo "[$(basename $lib)] $(sed -n 1p gpurun_out/ab.tmp | sed 's/nlml.*//') $(sed -n 2p gpurun_out/ab.tmp)"
  done
done
