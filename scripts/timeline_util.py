"""From a rocprofv3 kernel trace of scripts/prof_target.py: where in the last posterior is the chip NOT running a
large MFMA GEMM?  Prints the coverage per 2 ms bucket and the longest uncovered gaps."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rs_idx = [i for i, r in enumerate(rows) if "row_stats" in r["Kernel_Name"]]
end = rs_idx[-1]
cands = [i for i, r in enumerate(rows[:end]) if "se_build" in r["Kernel_Name"]]
gmax = max(int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) for i in cands)
start = [i for i in cands if int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) == gmax][-1]
seg = rows[start:end + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
print("window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
def wgs(r):
    return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
big = sorted([(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in seg if "gemm_nt8" in r["Kernel_Name"] and wgs(r) >= 400])
# union of intervals
un = []
for a, b in big:
    if un and a <= un[-1][1]:
        un[-1][1] = max(un[-1][1], b)
    else:
        un.append([a, b])
cov = sum(b - a for a, b in un)
print("large-GEMM coverage: %.2f ms of %.2f (%.1f%%); %d large launches" % (cov / 1e6, (t1 - t0) / 1e6, 100.0 * cov / (t1 - t0), len(big)))
gaps = []
prev = 0
for a, b in un:
    if a > prev: gaps.append((a - prev, prev, a))
    prev = b
if t1 - t0 > prev: gaps.append((t1 - t0 - prev, prev, t1 - t0))
gaps.sort(reverse=True)
print("longest uncovered gaps (ms: length @ start):", ", ".join("%.2f@%.1f" % (g / 1e6, s / 1e6) for g, s, e in gaps[:12]))
print("uncovered total %.2f ms in %d gaps; gaps < 50 us: %.2f ms" % (sum(g for g, _, _ in gaps) / 1e6, len(gaps), sum(g for g, _, _ in gaps if g < 50000) / 1e6))
B = 2e6
nb = int((t1 - t0) / B) + 1
line = []
for k in range(nb):
    lo, hi = k * B, (k + 1) * B
    c = sum(max(0, min(b, hi) - max(a, lo)) for a, b in un)
    line.append("%3d" % round(100 * c / B))
print("coverage % per 2 ms:", " ".join(line))
# what runs inside the biggest gaps
for g, s, e in gaps[:4]:
    names = {}
    for r in seg:
        a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        ov = min(b, e) - max(a, s)
        if ov > 0:
            k = r["Kernel_Name"].split("(")[0][:28] + "/%d" % wgs(r)
            names[k] = names.get(k, 0) + ov
    top = sorted(names.items(), key=lambda kv: -kv[1])[:5]
    print("  gap %.2f ms @ %.1f: " % (g / 1e6, s / 1e6) + ", ".join("%s %.0fus" % (k, v / 1e3) for k, v in top))
