#!/bin/bash
# kernel traces of three posteriors at N = 16384 in the stream layouts under comparison -> gpurun_out/trace_ab/*.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_direct -- python3 $R/scripts/prof_target.py 16384 8 1024 3 leaf_server=0 > $OUT/direct.log 2>&1 || exit 1
python3 $R/scripts/trace_ab.py $OUT/kt_direct > $OUT/direct.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_server -- python3 $R/scripts/prof_target.py 16384 8 1024 3 leaf_server=1 > $OUT/server.log 2>&1 || exit 1
python3 $R/scripts/trace_ab.py $OUT/kt_server > $OUT/server.txt 2>&1
export GPK_RESERVE_CU=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_masked -- python3 $R/scripts/prof_target.py 16384 8 1024 3 > $OUT/masked.log 2>&1 || exit 1
python3 $R/scripts/trace_ab.py $OUT/kt_masked > $OUT/masked.txt 2>&1
rm -rf $OUT/kt_direct $OUT/kt_server $OUT/kt_masked
