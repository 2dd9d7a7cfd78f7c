#!/bin/bash
# LDS counters of the trailing-update kernel (one rocprofv3 --pmc pass of the bench command): bank-conflict cycles against all
# LDS-array cycles, for the LDS-DMA form (gemm_pipe 2, default) and the register-staged plain loop (gemm_pipe 0).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pipe in 2 0; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p$pipe -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --opt gemm_pipe=$pipe > $OUT/p$pipe.log 2>&1 || exit 1
  python3 - $OUT/p$pipe $pipe <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if "gemm_nt8" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("gemm_pipe=%s gemm_nt8_kernel (all launches of 3 posteriors): " % sys.argv[2] + ", ".join("%s %.4g" % kv for kv in sorted(acc.items())) +
      " -> bank-conflict share of LDS-array cycles %.2f %%" % (100.0 * acc.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, acc.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
  rm -rf $OUT/p$pipe
done
