#!/bin/bash
# lazy corner (options corner_from / corner_chunk / corner_last / corner_prio): posterior time at N = 16384 over a grid, A/B against off
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in ${CFGS:-"0 2048 256 0 0" "10240 2048 256 0 7680" "10240 1024 256 0 7680" "10240 2560 256 0 6144" "11264 2048 256 0 7680" "12288 2048 256 0 7680" "9216 2048 256 0 7680" "10240 2048 256 0 8704"}; do
  set -- $cfg
  echo "== corner_from=$1 corner_chunk=$2 corner_last=$3 corner_prio=$4 corner_hold=$5 corner_from2=${6:-0} corner_from3=${7:-0} ${EXTRA}"
  timeout -k 10 120 python scripts/prof_target.py ${NN:-16384} ${DD:-8} 1024 10 timing=0 corner_from=$1 corner_chunk=$2 corner_last=$3 corner_prio=$4 corner_hold=$5 corner_from2=${6:-0} corner_from3=${7:-0} ${EXTRA} 2>/dev/null | sed -n 1,2p || exit 1
done
done
