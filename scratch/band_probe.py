"""Round 5 starting point of the banded kernels: BASELINE configs[3] with the band forced on the one-wave against the four-wave
tier, and the ONT-shaped grid (30 kbp @ 10 %, beta x lambda) against the exact run: kernel ms, speed-up, recall."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu

which = sys.argv[1] if len(sys.argv) > 1 else "all"
PEN = (2, 3, 1)


def run(al, batch, me, band, beta, cigar=True, reps=2):
    best = None
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d_scores, _ = al.align(batch, PEN, max_error=me, compute_cigar=cigar, band=band, band_width=beta, fetch=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        st = al.stats()
        rec = (dt, st.main_launch_ms, st.align_ms, st.trace_ms, list(st.pairs_tier), st.pairs_banded, st.blocks_per_cu_tier0, st.main_launch_cells)
        if best is None or dt < best[0]: best = rec
    return best, d_scores.cpu().numpy()


if which in ("all", "cfg4"):
    buf, meta = wfagpu.generate_pairs(16384, 10000, 0.03, seed=1000, nthreads=16)
    for mt in (1, 2, 3):
        al = wfagpu.DeviceAligner(0, force_band=1, band_tier=mt)
        batch = al.upload(buf, meta)
        for cigar in (True, False):
            r, s = run(al, batch, 3000, 25, 512, cigar=cigar)
            print(f"cfg4 band forced band_tier={mt} cigar={cigar}: step {r[0]:.2f} ms main {r[1]:.2f} align {r[2]:.2f} trace {r[3]:.2f} tiers {r[4]} banded {r[5]} bpc {r[6]} cells {r[7]/1e9:.2f} G", flush=True)
        al.close()
    al = wfagpu.DeviceAligner(0)
    batch = al.upload(buf, meta)
    r, s = run(al, batch, 3000, -1, 0)
    print(f"cfg4 exact: step {r[0]:.2f} ms main {r[1]:.2f} align {r[2]:.2f} trace {r[3]:.2f} tiers {r[4]} bpc {r[6]} cells {r[7]/1e9:.2f} G", flush=True)
    al.close()

if which in ("all", "ont"):
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    buf, meta = wfagpu.generate_pairs(n, 30000, 0.10, seed=1000, nthreads=16)
    al = wfagpu.DeviceAligner(0)
    batch = al.upload(buf, meta)
    r, s_exact = run(al, batch, 9000, -1, 0)
    ex_ms = r[0]
    print(f"ont exact: step {r[0]:.2f} ms main {r[1]:.2f} align {r[2]:.2f} trace {r[3]:.2f} tiers {r[4]} cells {r[7]/1e9:.2f} G mean score {s_exact.mean():.0f}", flush=True)
    al.close()
    tiers = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    for bt in tiers:
        al = wfagpu.DeviceAligner(0, force_band=1, band_tier=bt)
        batch = al.upload(buf, meta)
        print("band_tier", bt)
        for beta in (352, 512, 1024):
            for lam in ((10, 25, 50, 100, 750) if bt == 0 else (25, 750)):
                r, s = run(al, batch, 9000, lam, beta, reps=1)
                print(f"ont beta {beta} lambda {lam}: step {r[0]:.2f} ms main {r[1]:.2f} align {r[2]:.2f} trace {r[3]:.2f} tiers {r[4]} banded {r[5]} bpc {r[6]} cells {r[7]/1e9:.2f} G "
                      f"speedup {ex_ms / r[0]:.2f}x recall {(s == s_exact).mean() * 100:.2f} % excess {((s - s_exact) / np.maximum(s_exact, 1)).mean() * 100:.2f} %", flush=True)
        al.close()
