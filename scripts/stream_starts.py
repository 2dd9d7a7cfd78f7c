import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rs_idx = [i for i, r in enumerate(rows) if "row_stats" in r["Kernel_Name"]]
end = rs_idx[-1]
cands = [i for i, r in enumerate(rows[:end]) if "se_build" in r["Kernel_Name"]]
gmax = max(int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) for i in cands)
start = [i for i in cands if int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) == gmax][-1]
seg = rows[start:end + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
print("window %.3f ms" % ((t1 - t0) / 1e6))
for sid in sorted(set(r["Stream_Id"] for r in seg)):
    rs = [r for r in seg if r["Stream_Id"] == sid]
    print("   stream %s: n=%4d active %6.2f .. %6.2f ms (first: %s)" % (sid, len(rs), (int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (int(rs[-1]["End_Timestamp"]) - t0) / 1e6, rs[0]["Kernel_Name"].split("(")[0][:24]))
