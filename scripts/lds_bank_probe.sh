#!/bin/bash
# ds_read_b64 bank conflicts by address pattern, measured: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per probe kernel
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lds_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p -- python3 $R/scripts/lds_bank_probe.py > $OUT/run.log 2>&1 || exit 1
python3 - $OUT/p <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if "lds_bank_probe" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("<")[1].split(">")[0]
        acc.setdefault(k, {})[r["Counter_Name"]] = acc.setdefault(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = {"0": "lane * 8 (contiguous)", "1": "LDS-DMA image (128-B rows, 16-B chunks XOR (row>>1)&7)", "2": "rows 18 doubles apart", "3": "rows 17 doubles apart",
         "4": "128-B rows, 8-byte swizzle", "5": "128-B rows, no swizzle"}
for k in sorted(acc):
    a = acc[k]
    print("pattern %s %-58s LDS instr %.3g  IDX_ACTIVE %.4g  BANK_CONFLICT %.4g  -> %.1f cycles per wave-instruction, conflict share %.1f %%" % (
        k, names.get(k, ""), a.get("SQ_INSTS_LDS", 0), a.get("SQ_LDS_IDX_ACTIVE", 0), a.get("SQ_LDS_BANK_CONFLICT", 0),
        a.get("SQ_LDS_IDX_ACTIVE", 0) / max(1.0, a.get("SQ_INSTS_LDS", 1)), 100.0 * a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, a.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
rm -rf $OUT/p
