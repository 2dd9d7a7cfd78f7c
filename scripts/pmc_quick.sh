#!/bin/bash
# the two PMC passes only (HBM-side traffic of the trailing-update kernel), printed as one line
TAG=${1:-pmcq}
shift
OPTS="$@"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $OPTS > $OUT/pmc_$c.log 2>&1 || exit 1
done
python3 $R/scripts/pmc_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic_gemm_nt8.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python3 -c "
import json;d=json.load(open('$OUT/traffic_gemm_nt8.json'));print('$TAG: per launch %.3f GB (fetch x2 %.2f GB + write %.2f GB per factorisation), algorithmic %.3f GB -> %.2fx' % (d['hbm_bytes_per_launch']/1e9, d['fetch_bytes_corrected_x2']/1e9, d['write_bytes']/1e9, d['algorithmic_bytes_per_launch']/1e9, d['hbm_bytes_per_launch']/d['algorithmic_bytes_per_launch']))"
cd $R && python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $OPTS | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('%.2f posteriors/s, trailing %.1f TFLOP/s' % (d['value'], d['roofline']['achieved']))"
