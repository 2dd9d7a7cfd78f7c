#!/bin/bash
# instruction mix and wait cycles of the trailing-update kernel (one rocprofv3 --pmc pass each, of the bench command)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/s$i.log 2>&1 || exit 1
  python3 - $OUT/s$i <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if "gemm_nt8" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("  ".join("%s %.4g" % kv for kv in sorted(acc.items())))
PY
  rm -rf $OUT/s$i
done
