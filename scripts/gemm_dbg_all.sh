#!/bin/bash
# timing-only variants of the 8-wave kernel (library built with make DBG=1)
cd $GRAFT_REPO_ROOT
for d in ${DBGS:-0 8 1 9 2 16}; do
  GPK_GEMM_DBG=$d timeout -k 10 120 python scripts/gemm_dbg.py 7936 2048 512 256 128 || exit 1
done
