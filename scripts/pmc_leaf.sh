#!/bin/bash
# LDS counters of the leaf kernel (scripts/leaf_phases.py under rocprofv3 --pmc), per timing-only variant
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_leaf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p -- python3 $R/scripts/leaf_phases.py > $OUT/run.log 2>&1 || exit 1
python3 - $OUT/p <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "potf2" in r["Kernel_Name"]]
# group launches by dispatch order into the 7 variants x 2 runs x 401 launches of leaf_phases.py: use Dispatch_Id order
byd = {}
for r in rows:
    byd.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(byd)
per = 401
names = ["full leaf", "no elimination", "no inverse", "neither", "no panel solve inside", "no trailing tile updates", "no diagonal blocks"]
for v in range(len(ids) // (2 * per)):
    seg = ids[(2 * v + 1) * per:(2 * v + 2) * per]
    acc = {}
    for i in seg:
        for k, x in byd[i].items(): acc[k] = acc.get(k, 0.0) + x
    n = len(seg)
    print("%-26s per launch: LDS instr %.0f  LDS-array cycles %.0f  bank-conflict cycles %.0f (%.0f %%)  wave cycles waiting on LDS %.0f of %.0f  VALU instr %.0f" % (
        names[v] if v < len(names) else v, acc["SQ_INSTS_LDS"] / n, acc["SQ_LDS_IDX_ACTIVE"] / n, acc["SQ_LDS_BANK_CONFLICT"] / n,
        100.0 * acc["SQ_LDS_BANK_CONFLICT"] / max(1.0, acc["SQ_LDS_IDX_ACTIVE"]), acc["SQ_WAIT_INST_LDS"] / n, acc["SQ_WAVE_CYCLES"] / n, acc["SQ_INSTS_VALU"] / n))
PY
rm -rf $OUT/p
