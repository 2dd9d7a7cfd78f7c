#!/bin/bash
# rocprofv3 kernel trace of a few posteriors at one size; keeps the LAST posterior's rows as a compact CSV under gpurun_out/
# usage: trace_dump.sh N d tag [options...]
R=$GRAFT_REPO_ROOT
N=$1; D=$2; TAG=$3; shift 3
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/scripts/prof_target.py $N $D 1024 3 timing=0 "$@" > $OUT/run.log 2>&1 || exit 1
python3 - "$OUT" <<'P'
import csv, glob, sys
out = sys.argv[1]
f = sorted(glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rs = [i for i, r in enumerate(rows) if "row_stats" in r["Kernel_Name"]]
seg = rows[rs[-2] + 1:rs[-1] + 1]
t0 = min(int(r["Start_Timestamp"]) for r in seg)
with open(out + "/last_posterior.csv", "w") as g:
    g.write("stream,start_us,end_us,wgs,kernel\n")
    for r in seg:
        wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        g.write("%s,%.2f,%.2f,%d,%s\n" % (r["Stream_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, wg, k))
print(len(seg), "kernels")
P
rm -rf $OUT/kt
