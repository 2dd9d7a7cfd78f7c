"""From a rocprofv3 kernel trace of scripts/prof_target.py: per-stream busy time of the LAST posterior in 2 ms buckets,
plus per-stream totals by kernel.  Usage: stream_timeline.py <trace dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rs = [i for i, r in enumerate(rows) if "row_stats" in r["Kernel_Name"]]
seg = rows[rs[-2] + 1:rs[-1] + 1]
t0 = min(int(r["Start_Timestamp"]) for r in seg)
t1 = max(int(r["End_Timestamp"]) for r in seg)
print("window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
def wgs(r): return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
streams = sorted(set(r["Stream_Id"] for r in seg))
B = 2e6
nb = int((t1 - t0) / B) + 1
for sid in streams:
    ks = [r for r in seg if r["Stream_Id"] == sid]
    iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in ks)
    un = []
    for a, b in iv:
        if un and a <= un[-1][1]: un[-1][1] = max(un[-1][1], b)
        else: un.append([a, b])
    busy = sum(b - a for a, b in un)
    line = []
    for k in range(nb):
        lo, hi = k * B, (k + 1) * B
        c = sum(max(0, min(b, hi) - max(a, lo)) for a, b in un)
        line.append("%3d" % round(100 * c / B))
    names = {}
    for r in ks:
        k = r["Kernel_Name"].split("(")[0].split("<")[0][-20:]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = names.setdefault(k, [0, 0]); e[0] += d; e[1] += 1
    top = sorted(names.items(), key=lambda kv: -kv[1][0])[:6]
    print("stream %s: %d kernels, busy %.2f ms" % (sid, len(ks), busy / 1e6))
    print("   busy %% per 2 ms: " + " ".join(line))
    print("   " + ", ".join("%s %.2fms/%d" % (k, v[0] / 1e6, v[1]) for k, v in top))
# the big launches on any stream, in time order
print("large launches (>= 400 workgroups, > 0.3 ms):")
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if wgs(r) >= 400 and d > 0.3:
        print("   s%s %-18s wgs=%6d start %7.3f dur %6.3f ms" % (r["Stream_Id"], r["Kernel_Name"].split("(")[0].split("<")[0][-18:], wgs(r), (int(r["Start_Timestamp"]) - t0) / 1e6, d))
