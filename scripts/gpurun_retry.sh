#!/bin/bash
# local helper: run one gpurun call, retrying ONLY while no GPU slot is free (exit code 3: nothing ran, nothing charged)
# usage: scripts/gpurun_retry.sh <timeout> '<command>'
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
