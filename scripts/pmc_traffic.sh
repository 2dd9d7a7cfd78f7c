#!/bin/bash
# HBM traffic of the dominant kernel (gemm_nt8_kernel) from PMC counters: two separate passes
# (FETCH_SIZE, WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  tail -1 gpurun_out/pmc_$c.log | cut -c1-200
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
tail -1 gpurun_out/prof_bench.log | cut -c1-300
# timing-event overhead check
timeout -k 10 100 python scripts/prof_target.py 16384 8 1024 3 | head -3
