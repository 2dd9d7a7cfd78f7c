"""From the rocprofv3 kernel trace of `bench.py`: the trailing-update launches only (gemm_nt8_kernel on the update
stream), i.e. the launches bench.py brackets with HIP events for its roofline object.  Prints count, average and
total duration so the trace can be checked against roofline.avg_launch_ms."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
def wgs(r): return int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]))
kb = [r for r in rows if "se_build" in r["Kernel_Name"]]
main = max(kb, key=wgs)["Stream_Id"]          # the K(X,X) build runs on the update stream
tr = [r for r in rows if "gemm_nt8" in r["Kernel_Name"] and r["Stream_Id"] == main]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
allg = [r for r in rows if "gemm_nt8" in r["Kernel_Name"]]
print("kernel trace: %s" % f.split("/")[-1])
print("gemm_nt8_kernel launches, all streams: %d (avg %.4f ms) -- includes the prediction and look-ahead GEMMs" % (len(allg), sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in allg) / len(allg)))
print("gemm_nt8_kernel launches on the update stream (stream %s) = trailing updates: %d, avg %.4f ms, total %.2f ms" % (main, len(dur), sum(dur) / len(dur), sum(dur)))
