#!/bin/bash
# chain-stream GEMMs at raised wave priority (GPK_CHAIN_PRIO) x the 64x64-tile kernel in 32 KiB of LDS (GPK_SMALL_LDS)
cd $GRAFT_REPO_ROOT
GPK_SMALL_LDS=1 GPK_CHAIN_PRIO=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or bitwise or round_identically or two_segment or potrf" 2>&1 | tail -2
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== chain_prio=$1 small_lds=$2"
  for nd in "16384 8" "8192 8" "4096 4"; do
    GPK_CHAIN_PRIO=$1 GPK_SMALL_LDS=$2 timeout -k 10 120 python scripts/prof_target.py $nd 1024 10 timing=0 2>/dev/null | sed -n 1,1p || exit 1
  done
done
done
