"""From gpurun_out/trace_<tag>/last_posterior.csv: the 64x64-tile GEMM launches by stream and grid size (launches, mean, total), and the
same for the window after a given time -- which skinny GEMMs does the chain-bound part spend its stream time in?"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
agg = collections.OrderedDict()
for r in rows:
    if "gemm_nt_kernel" not in r["kernel"] or float(r["start_us"]) < t_from: continue
    w = int(r["wgs"])
    b = "<=16" if w <= 16 else "<=64" if w <= 64 else "<=128" if w <= 128 else "<=256" if w <= 256 else "<=512" if w <= 512 else ">512"
    e = agg.setdefault((r["stream"], b), [0, 0.0, 0.0])
    d = float(r["end_us"]) - float(r["start_us"])
    e[0] += 1; e[1] += d; e[2] = max(e[2], d)
print("64x64-tile GEMM launches from %.0f us on, by stream and workgroup count:" % t_from)
for (s, b), (n, t, mx) in sorted(agg.items()):
    print("  stream %s wgs %-6s n=%4d mean %6.1f us max %6.1f total %8.1f us" % (s, b, n, t / n, mx, t))
