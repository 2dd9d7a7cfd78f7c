"""MFMA-pipe utilisation of the trailing-update kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_F64, GRBM_GUI_ACTIVE) of `bench.py --steps 2 --warmup 1`: the gemm_nt8_kernel launches of the
LAST posterior on the update stream (the ones bench.py brackets with HIP events).
    MfmaUtil   = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * 1024 SIMDs)      (rocprofv3's own derived formula)
    flops      = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512                                     (counts executed MFMA flops)"""
import csv, glob, json, sys

f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
rows = list(csv.DictReader(open(f)))

def wgs(r):
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return g // w

by = {}
dur = {}
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"], r["Queue_Id"], wgs(r))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    dur[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
disp = sorted(by.items(), key=lambda kv: int(kv[0][0]))
kb = [k for k, _ in disp if "se_build" in k[1]]
main_q = max(kb, key=lambda k: k[3])[2]
tr = [(k, v) for k, v in disp if "gemm_nt8" in k[1] and k[2] == main_q]
per_step = len(tr) // 3
tr = tr[-per_step:]
busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for _, v in tr)
act = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for _, v in tr)
mops = sum(v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) for _, v in tr)
t_ns = sum(dur[k] for k, _ in tr)
NXCD, NSIMD = 8, 1024            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (rocprofv3's MfmaUtil takes the max instead)
cyc = act / NXCD
print(json.dumps({
    "kernel": "gemm_nt8_kernel (trailing-update launches of one factorisation on the update stream, N=16384)",
    "launches": len(tr),
    "SQ_VALU_MFMA_BUSY_CYCLES_sum": busy, "GRBM_GUI_ACTIVE_sum": act, "SQ_INSTS_VALU_MFMA_MOPS_F64_sum": mops,
    "kernel_time_ms_sum": t_ns / 1e6,
    "shader_clock_ghz": cyc / t_ns if t_ns else None,
    "mfma_util_percent": 100.0 * busy / (cyc * NSIMD) if cyc else None,
    "mfma_flops_executed": mops * 512.0,
    "tflops_in_this_pass": mops * 512.0 / (t_ns * 1e-9) / 1e12 if t_ns else None,
    "note": "MfmaUtil = busy cycles of the matrix pipes (= 64 per v_mfma_f64_16x16x4) over GPU-active cycles x 1024 SIMDs.  Counter "
            "collection SERIALISES dispatches: in this pass the launches run alone, not next to the panel chain and the pipelined "
            "prediction as in the timed bench.  Executed flops include the full diagonal tiles (the algorithmic count m^2 w takes half "
            "of them): the counter confirms the flop count bench.py uses to within that 1 %.",
}, indent=1))
