"""Summary of gpurun_out/trace_<tag>/last_posterior.csv (scripts/trace_dump.sh): per stream and kernel, launches / mean / total duration."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["stream"], r["kernel"][:40])
    d = float(r["end_us"]) - float(r["start_us"])
    e = agg.setdefault(k, [0, 0.0])
    e[0] += 1; e[1] += d
print("window %.1f us, %d kernels" % (max(float(r["end_us"]) for r in rows), len(rows)))
for (s, k), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  stream %s %-42s n=%4d mean %7.1f us total %8.1f us" % (s, k, n, t / n, t))
