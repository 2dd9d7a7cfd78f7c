"""Per-launch HBM traffic of the trailing-update kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 2 --warmup 1`: the gemm_nt8_kernel launches of the LAST posterior that ran on the update stream
(= the launches bench.py brackets with HIP events).  Units and the gfx950 correction follow MI355X_MICROARCH.md
(counters in KiB; FETCH_SIZE doubled)."""
import csv, glob, json, sys

def load(d, counter):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"))[-1]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    return rows

def wgs(r):
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    return g // w

def trailing(rows):
    qkey = "Queue_Id"
    kb = [r for r in rows if "se_build" in r["Kernel_Name"]]
    main_q = max(kb, key=wgs)[qkey]
    g = [r for r in rows if "gemm_nt8" in r["Kernel_Name"] and r[qkey] == main_q]
    per_step = len(g) // 3            # warmup 1 + steps 2
    return g[-per_step:]

def alg_model(N=16384, nb_big_until=8192, nb_small_from=4096, tail_rows=4096, small_tile_below=256):
    """(bytes, flops) of every trailing-update launch of the 8-wave kernel in one factorisation, from the default panel
    schedule (same rules as gpk_core.hip: model_create / potrf_device / potrf_tail)."""
    c0 = 0; sched = []
    while c0 < N:
        rem = N - c0
        w = 512 if rem > nb_big_until + 512 else (128 if rem <= nb_small_from else 256)
        sched.append((c0, w)); c0 += w
    tail = [i for i, (c, w) in enumerate(sched) if w == 128 and N - c <= tail_rows]
    t0 = tail[0] if len(tail) >= 4 else len(sched)      # first panel of the tile pipeline
    ms = []
    for p, (c, w) in enumerate(sched):
        if p < t0:
            if p + 1 == t0: m = N - (c + w) - 384        # hand-over: three tile columns stay on the chain stream
            else: m = N - c - w - (sched[p + 1][1] if p + 1 < len(sched) else 0)
        else:
            m = N - c - 512                              # REST(p) of the tile pipeline
        ms.append((m, w))
    out = []
    for m, w in ms:
        if m <= 0: continue
        nt = m // 128
        if nt * (nt + 1) // 2 < small_tile_below: continue   # small launches use the 64x64-tile kernel
        out.append((16.0 * 128 * 128 * nt * (nt + 1) / 2 + 8.0 * m * w, float(m) * m * w))
    return out


def main():
    fe = trailing(load(sys.argv[1], "FETCH_SIZE"))
    wr = trailing(load(sys.argv[2], "WRITE_SIZE"))
    assert len(fe) == len(wr) and len(fe) > 0, (len(fe), len(wr))
    fetch_raw = sum(float(r["Counter_Value"]) for r in fe)
    write_raw = sum(float(r["Counter_Value"]) for r in wr)
    n = len(fe)
    # algorithmic bytes: every lower 128x128 tile of C read + written once (16 B per element) + the panel read once;
    # exact algorithmic figures come from the panel schedule (same rule as gpk_core.hip model_create / potrf_device)
    alg_list = alg_model()
    out = {
        "kernel": "gemm_nt8_kernel (trailing-update launches of one factorisation on the update stream, N=16384)",
        "launches": n,
        "launches_expected_from_schedule": len(alg_list),
        "fetch_size_kb_raw": fetch_raw,
        "write_size_kb_raw": write_raw,
        "fetch_bytes_corrected_x2": fetch_raw * 1024.0 * 2.0,
        "write_bytes": write_raw * 1024.0,
        "hbm_bytes_per_launch": (fetch_raw * 1024.0 * 2.0 + write_raw * 1024.0) / n,
        "algorithmic_bytes_per_launch": sum(a for a, _ in alg_list) / max(1, len(alg_list)),
        "algorithmic_flops_per_launch": sum(f for _, f in alg_list) / max(1, len(alg_list)),
        "note": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --steps 2 --warmup 1`; counters in KiB; "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); algorithmic = 16 B per element of the lower C tiles (read + written once) "
                "+ 8*m*w (panel read once)",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
