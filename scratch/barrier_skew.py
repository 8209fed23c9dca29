"""Per-score barrier arrival skew in the multi-wave tiers (tuning.timed_barriers): workgroup 0 records s_memtime when each of its
waves reaches the per-score barrier and when it leaves it.  For cfg4 (four waves per alignment, tier 1) and cfg5 (sixteen
waves, hybrid ring, tier 4): how much of a score's period do the waves spend parked at the barrier, and is that the wait for
ONE slow wave (skew) or do they arrive together?  -> gpurun_out/barrier_skew.md"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
out = ["# Barrier arrival skew of the multi-wave tiers (s_memtime, workgroup 0, every score of its alignments)", "",
       "period = release of score s to release of score s+1; busy = release of the previous score to a wave's arrival; parked = arrival to release.",
       "skew = last arrival - first arrival of a score.  All in shader clock ticks of s_memtime (100 MHz constant clock on gfx950: 1 tick = 10 ns).", ""]
for name, n, L, err, me in (("cfg4 (16k x 10 kbp @ 3 %, tier 1: 4 waves)", 16384, 10000, 0.03, 3000), ("cfg5 (1024 x 30 kbp @ 10 %, tier 4: 16 waves, hybrid ring)", 1024, 30000, 0.10, 9000)):
    buf, meta = wfagpu.generate_pairs(n, L, err, seed=1000, nthreads=16)
    al = wfagpu.DeviceAligner(0, timed_barriers=1, no_auto_budget=1)
    batch = al.upload(buf, meta)
    al.align(batch, (2, 3, 1), max_error=me, compute_cigar=True, fetch=False)
    al.align(batch, (2, 3, 1), max_error=me, compute_cigar=True, fetch=False)
    st = al.stats()
    ptr, rec, waves = C.c_void_p(), C.c_uint(), C.c_int()
    al.lib.wfagpu_amd_debug_times.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    al.lib.wfagpu_amd_debug_times(al.ctx, C.byref(ptr), C.byref(rec), C.byref(waves))
    nw = waves.value
    raw = wfagpu._d2h(ptr.value, rec.value * nw * 24).view(np.uint64).reshape(rec.value, nw, 3)
    valid = raw[:, 0, 0] != 0
    r = raw[valid]
    arr, rel = r[:, :, 0].astype(np.int64), r[:, :, 1].astype(np.int64)
    score = (r[:, 0, 2] >> np.uint64(32)).astype(np.int64); width = (r[:, 0, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64) + 1
    # consecutive scores of one alignment: score increases by one
    cont = np.nonzero(np.diff(score) == 1)[0] + 1
    period = rel[cont].max(axis=1) - rel[cont - 1].max(axis=1)
    busy = arr[cont] - rel[cont - 1].max(axis=1)[:, None]
    parked = rel[cont] - arr[cont]
    skew = arr[cont].max(axis=1) - arr[cont].min(axis=1)
    last = arr[cont].argmax(axis=1)
    w = width[cont]
    out += [f"## {name}", "", f"main launch {st.main_launch_ms:.2f} ms (timed instantiation), tier {st.main_launch_tier}; {len(cont)} scores recorded, {nw} waves", "",
            f"| wavefront width | scores | period | busy (mean of waves) | busy (slowest wave) | parked (mean of waves) | skew | parked share of the period |", "|---|---|---|---|---|---|---|---|"]
    edges = [0, 256, 512, 1024, 2048, 4096, 1 << 30]
    for a, b in zip(edges[:-1], edges[1:]):
        m = (w >= a) & (w < b)
        if m.sum() < 10: continue
        out.append(f"| {a}-{min(b, int(w.max()) + 1) - 1} | {int(m.sum())} | {period[m].mean():.1f} | {busy[m].mean():.1f} | {busy[m].max(axis=1).mean():.1f} | {parked[m].mean():.1f} | {skew[m].mean():.1f} | {parked[m].mean() / period[m].mean() * 100:.1f} % |")
    out.append(f"| all | {len(cont)} | {period.mean():.1f} | {busy.mean():.1f} | {busy.max(axis=1).mean():.1f} | {parked.mean():.1f} | {skew.mean():.1f} | {parked.mean() / period.mean() * 100:.1f} % |")
    hist = np.bincount(last, minlength=nw)
    out += ["", "which wave arrives last (share of scores): " + ", ".join(f"w{i} {h / len(last) * 100:.0f} %" for i, h in enumerate(hist)), ""]
    q = np.percentile(skew / np.maximum(period, 1), [10, 50, 90])
    out += [f"skew / period: 10th percentile {q[0]:.2f}, median {q[1]:.2f}, 90th {q[2]:.2f}", ""]
    al.close(); del batch
open(os.path.join(ROOT, "gpurun_out", "barrier_skew.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
