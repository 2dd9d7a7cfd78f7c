#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/scripts/prof_target.py 16384 8 1024 3 > $OUT/run.log 2>&1 || exit 1
python3 $R/scripts/stream_timeline.py $OUT/kt > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
