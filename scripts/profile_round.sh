#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command, the trailing
# launches filtered from the same trace, and the two PMC passes for the HBM traffic of the trailing-update kernel.
# usage: scripts/profile_round.sh r02     -> gpurun_out/<tag>/...  (copy what is to be judged into profiles/)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-clock-probe --no-phases --no-secondary > $OUT/bench_line_under_rocprof.json 2> $OUT/kt.log || exit 1
cp $OUT/kt/kt_kernel_stats.csv $OUT/bench_kernel_stats.csv
python3 $R/scripts/trailing_from_trace.py $OUT > $OUT/bench_trailing_launches.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-phases --no-secondary > $OUT/pmc_$c.log 2>&1 || exit 1
done
python3 $R/scripts/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic_gemm_nt8.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_MFMA -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-phases --no-secondary > $OUT/pmc_MFMA.log 2>&1 || exit 1
python3 $R/scripts/pmc_mfma_json.py $OUT/pmc_MFMA > $OUT/mfma_util_gemm_nt8.json
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_MFMA
# the un-profiled bench line LAST, and with this run's PMC JSONs in place under profiles/ first: bench.py embeds the newest
# profiles/r*_traffic_gemm_nt8.json / r*_mfma_util_gemm_nt8.json, so the three files of one round agree on launches, traffic and MfmaUtil
mkdir -p $R/profiles
cp $OUT/traffic_gemm_nt8.json $R/profiles/${TAG}_traffic_gemm_nt8.json
cp $OUT/mfma_util_gemm_nt8.json $R/profiles/${TAG}_mfma_util_gemm_nt8.json
cd $R && python3 bench.py --steps 25 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
for f in bench_kernel_stats.csv bench_trailing_launches.txt bench_line_under_rocprof.json bench_line.json traffic_gemm_nt8.json mfma_util_gemm_nt8.json; do cp $OUT/$f $OUT/${TAG}_$f; done
ls -la $OUT
