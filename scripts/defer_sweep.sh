#!/bin/bash
# pipelined prediction: far-column updates of the early panels deferred (option pred_defer_from)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in ${CFGS:-"0" "-1" "8192" "10240" "12288"}; do
  set -- $cfg
  echo "== pred_defer_from=$1"
  timeout -k 10 120 python scripts/prof_target.py ${NN:-16384} ${DD:-8} 1024 10 timing=0 pred_defer_from=$1 2>/dev/null | sed -n 1,2p || exit 1
done
done
