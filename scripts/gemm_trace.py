"""Where a tile's fixed cost goes: per-workgroup stamps of ONE launch of the 8-wave GEMM (library built with `make DBG=1`).
   GPK_GEMM_DBG=32 GPK_GEMM_TRACE=/tmp/t.bin python scripts/gemm_trace.py m n k
Prints, in microseconds: prologue (kernel entry -> first MFMA: tile decode, first two slabs by DMA, C tile), main loop, epilogue
(stores issued), and per CU the gap between one workgroup's exit and the entry of the next one in its slot, plus how much of
every workgroup's non-MFMA time is covered by the MFMA phase of another workgroup on the same CU."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gptest_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpk_hooks import hooks  # libgpk_hooks.so: probes and test hooks live outside the product library

m, n, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (7936, 8192, 512)
path = os.environ.get("GPK_GEMM_TRACE", "/tmp/gemm_trace.bin")
os.environ["GPK_GEMM_TRACE"] = path
os.environ.setdefault("GPK_GEMM_DBG", "32")
_lib.init()
ms = ctypes.c_double()
rc = hooks().gpk_bench_gemm_nt(m, n, k, 0, 128, 3, ctypes.byref(ms))
print("m=%d n=%d k=%d: rc=%d %.3f ms per launch, %.2f TFLOP/s" % (m, n, k, rc, ms.value, 2.0 * m * n * k / ms.value / 1e9))
r = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
r = r[r[:, 0] > 0]
t0 = r[:, 0].min()
ent, fst, lst, ext = [(r[:, i] - t0).astype(float) * 0.01 for i in range(4)]
cu = r[:, 4].astype(int)
print("workgroups %d on %d CUs; launch spans %.1f us" % (len(r), len(set(cu)), ext.max()))
print("prologue (entry -> first MFMA)  mean %.2f  p50 %.2f  p90 %.2f us" % ((fst - ent).mean(), np.median(fst - ent), np.percentile(fst - ent, 90)))
sa, sb, sc = [(r[:, i] - t0).astype(float) * 0.01 for i in (5, 6, 7)]
print("   of it: entry -> set-up done %.2f | DMA + C loads issued %.2f | loads landed (wave 0) %.2f | first barrier %.2f us" % ((sa - ent).mean(), (sb - sa).mean(), (sc - sb).mean(), (fst - sc).mean()))
print("main loop                       mean %.2f  p50 %.2f  p90 %.2f us" % ((lst - fst).mean(), np.median(lst - fst), np.percentile(lst - fst, 90)))
print("epilogue (stores issued)        mean %.2f  p50 %.2f us" % ((ext - lst).mean(), np.median(ext - lst)))
gaps, cover, per_cu = [], [], []
for c in set(cu):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(ent[idx])]
    per_cu.append(len(idx))
    # slot turnover: every exit is matched with the first later entry on this CU
    e_sorted = np.sort(ext[idx])
    for x in e_sorted:
        later = ent[idx][ent[idx] >= x]
        if len(later):
            gaps.append(later.min() - x)
    # how much of each workgroup's [last MFMA, next first MFMA in the same slot] window is covered by another workgroup's main loop
    for i in idx:
        lo, hi = lst[i], lst[i] + 1e9
        later = idx[ent[idx] >= ext[i]]
        if not len(later):
            continue
        j = later[np.argmin(ent[later])]
        hi = fst[j]
        others = [o for o in idx if o != i and o != j]
        cov = sum(max(0.0, min(hi, lst[o]) - max(lo, fst[o])) for o in others)
        cover.append((hi - lo, min(cov, hi - lo)))
gaps = np.array(gaps)
print("workgroups per CU: min %d max %d" % (min(per_cu), max(per_cu)))
print("slot turnover (exit -> next entry on the same CU): mean %.2f p50 %.2f p90 %.2f us" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
w = np.array(cover)
print("non-MFMA window per tile (last MFMA of one tile -> first MFMA of the next in that slot): mean %.2f us, of which %.2f us (%.0f %%) lie under another workgroup's main loop on the same CU"
      % (w[:, 0].mean(), w[:, 1].mean(), 100 * w[:, 1].sum() / w[:, 0].sum()))

# does a workgroup speed up while its CU-mate is out of its main loop?  D = main-loop duration, solo = the part of it during
# which no other workgroup of the CU is between ITS first and last MFMA.  If a lone workgroup ran at the pair's rate, every
# solo microsecond would shorten D by half a microsecond: slope of D on solo = -0.5 (and -1/3 ... for 3 per CU).
D, S = [], []
for c in set(cu):
    idx = np.where(cu == c)[0]
    for i in idx:
        others = [o for o in idx if o != i]
        ev = sorted([(max(fst[i], fst[o]), min(lst[i], lst[o])) for o in others if min(lst[i], lst[o]) > max(fst[i], fst[o])])
        covered, cur = 0.0, fst[i]
        for a, b in ev:
            a = max(a, cur)
            if b > a:
                covered += b - a
                cur = b
        D.append(lst[i] - fst[i])
        S.append((lst[i] - fst[i]) - covered)
D, S = np.array(D), np.array(S)
A = np.vstack([np.ones_like(S), S]).T
coef = np.linalg.lstsq(A, D, rcond=None)[0]
print("main loop D against its solo part S: D = %.2f %+.3f * S us (mean S %.2f, corr %.2f)" % (coef[0], coef[1], S.mean(), np.corrcoef(D, S)[0, 1]))
for lo_, hi_ in ((0, 2), (2, 6), (6, 12), (12, 25), (25, 1e9)):
    sel = (S >= lo_) & (S < hi_)
    if sel.sum():
        print("   solo %4.0f..%-4.0f us: %5d workgroups, main loop mean %.2f us" % (lo_, min(hi_, 999), sel.sum(), D[sel].mean()))
if int(os.environ.get("GPK_GEMM_DBG", "32")) & 512:
    cyc = r[:, 5].astype(float)
    ghz = cyc / ((lst - fst) * 1e3)
    print("shader clock over the main loops (s_memtime / s_memrealtime): mean %.3f GHz, p10 %.3f, p90 %.3f" % (ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 90)))
if os.environ.get("GPK_TRACE_SAVE"):
    np.save(os.environ["GPK_TRACE_SAVE"], r)
_lib.shutdown()
