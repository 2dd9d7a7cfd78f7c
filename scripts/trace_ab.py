"""From a rocprofv3 kernel trace of scripts/prof_target.py: the LAST posterior's trailing-update launches one by one and
the panel chain's leaf statistics (time from the end of the previous kernel on the chain stream to the start of the
leaf, leaf duration).  Usage: trace_ab.py <trace dir>"""
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
kb = [r for r in seg if "se_build" in r["Kernel_Name"]]
main = max(kb, key=wgs)["Stream_Id"]
tr = [r for r in seg if "gemm_nt8" in r["Kernel_Name"] and r["Stream_Id"] == main]
print("trailing launches on stream %s: %d, total %.2f ms" % (main, len(tr), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) / 1e6))
line = []
for r in tr:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    line.append("%d:%.0fus(%.1fus/kWG)" % (wgs(r), d, d / wgs(r) * 1e3))
print("  " + "  ".join(line))
leaf = [r for r in seg if "potf2" in r["Kernel_Name"] or "leaf_proxy" in r["Kernel_Name"]]
if leaf:
    chain = leaf[0]["Stream_Id"]
    ck = [r for r in seg if r["Stream_Id"] == chain]
    gaps, durs, gaps_other, durs_other = [], [], [], []
    for a, b in zip(ck[:-1], ck[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        d = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
        if "potf2" in b["Kernel_Name"] or "leaf_proxy" in b["Kernel_Name"]:
            gaps.append(g); durs.append(d)
        else:
            gaps_other.append(g); durs_other.append(d)
    q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]
    print("chain stream %s: %d kernels, busy %.2f ms" % (chain, len(ck), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ck) / 1e6))
    print("  leaves: %d, duration avg %.1f us (median %.1f, p90 %.1f, max %.1f); gap before: avg %.1f us (median %.1f, p90 %.1f, max %.1f)" % (len(durs), sum(durs) / len(durs), q(durs, .5), q(durs, .9), max(durs), sum(gaps) / len(gaps), q(gaps, .5), q(gaps, .9), max(gaps)))
    print("  other chain kernels: %d, duration avg %.1f us (median %.1f, p90 %.1f); gap before: avg %.1f us (median %.1f, p90 %.1f, max %.1f), total gaps %.2f ms" % (len(durs_other), sum(durs_other) / len(durs_other), q(durs_other, .5), q(durs_other, .9), sum(gaps_other) / len(gaps_other), q(gaps_other, .5), q(gaps_other, .9), max(gaps_other), sum(gaps_other) / 1e3))
    # second half of the columns: leaves in the last half of the window
    print("  leaf durations in order: " + " ".join("%.0f" % d for d in durs))
    print("  gaps before leaves in order: " + " ".join("%.0f" % g for g in gaps))
