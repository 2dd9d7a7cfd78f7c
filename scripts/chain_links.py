"""From a rocprofv3 kernel trace of scripts/prof_target.py: the kernels of the panel chain (panel + leaf streams)
in two windows of the last posterior, with durations and gaps."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rs = [i for i, r in enumerate(rows) if "row_stats" in r["Kernel_Name"]]
seg = rows[rs[-2] + 1:rs[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
def wgs(r): return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
leaf_streams = set(r["Stream_Id"] for r in seg if "potf2" in r["Kernel_Name"])
# the panel stream = the stream of the kernel that follows a leaf most often
nxt = {}
for a, b in zip(seg, seg[1:]):
    pass
ch_streams = set(leaf_streams)
for sid in set(r["Stream_Id"] for r in seg):
    ks = [r for r in seg if r["Stream_Id"] == sid]
    if any("gemm_nt_kernel" in r["Kernel_Name"] and wgs(r) < 600 for r in ks) and not any("se_build" in r["Kernel_Name"] or "diag_solve" in r["Kernel_Name"] for r in ks):
        ch_streams.add(sid)
ch = [r for r in seg if r["Stream_Id"] in ch_streams]
for lo, hi in [tuple(float(x) for x in w.split("-")) for w in sys.argv[2:]] or [(29.3, 30.3), (34.9, 35.5)]:
    print("--- %.1f..%.1f ms (streams %s)" % (lo, hi, sorted(ch_streams)))
    prev = None
    for r in ch:
        a = (int(r["Start_Timestamp"]) - t0) / 1e6; b = (int(r["End_Timestamp"]) - t0) / 1e6
        if lo <= a <= hi:
            print("  s%s %-22s wgs=%4d start %.3f dur %5.1f us  gap %5.1f us" % (r["Stream_Id"], r["Kernel_Name"].split("(")[0][-22:], wgs(r), a, (b - a) * 1e3, (a - prev) * 1e3 if prev else 0))
            prev = b
