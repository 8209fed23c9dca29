"""Throughput of the exact kernels under other penalty sets, resident batch: 1M x 1 kbp @ 5 % (score + CIGAR) and 100k x 150 bp @ 2 %
(score-only and with CIGARs).  Prints the step, the main launch, its tier and its cells/s (profiles/r06/penalty_sets*.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "all"
reps = 3
def leg(tag, buf, meta, sets, cigar, k=500):
    al = wfagpu.DeviceAligner(0); batch = al.upload(buf, meta)
    n = len(meta)
    for pen, me in sets:
        al.hint_same_stream(False)
        al.align(batch, pen, max_error=me, compute_cigar=cigar, fetch=False)
        al.hint_same_stream(True)
        al.align(batch, pen, max_error=me, compute_cigar=cigar, fetch=False)
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d_s, ptrs = al.align(batch, pen, max_error=me, compute_cigar=cigar, fetch=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            st = al.stats()
            if best is None or dt < best[0]: best = (dt, st.align_ms, st.trace_ms, st.main_launch_ms, st.main_launch_tier, st.main_launch_cells, st.cells, st.auto_budget, st.pairs_budget_missed)
        so, co, _ = oracle_lib.oracle_batch(buf, meta[:k], pen, cigar=cigar, nthreads=16)
        ok = np.array_equal(d_s[:k].cpu().numpy(), so)
        if cigar:
            cg = wfagpu.fetch_cigars(ptrs[0], ptrs[1], ptrs[2], n, st.text_bytes)[:k]
            ok = ok and cg == co
        dt, a, t, m, tier, mc, c, ab, miss = best
        print(f"{tag} penalties {pen} -e {me}: step {dt*1e3:.3f} ms  {n/dt/1e6:.2f} M/s  align {a:.3f} trace {t:.3f}  main launch {m:.3f} ms tier {tier} "
              f"cells {mc/1e9:.3f} G -> {mc/m/1e6 if m else 0:.1f} G cells/s  (call cells {c/1e9:.3f} G, budget {ab}, missed {miss})  exact {ok}", flush=True)
    al.close()
if which in ("all", "1k"):
    buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
    leg("1kbp+cigar", buf, meta, (((2, 3, 1), 300), ((5, 3, 2), 600), ((3, 1, 4), 700), ((4, 6, 2), 600), ((1, 2, 1), 200), ((7, 2, 3), 900)), True)
if which in ("all", "150"):
    buf, meta = wfagpu.generate_pairs(max(n // 10, 1000), 150, 0.02, seed=1000, nthreads=16)
    sets = (((2, 3, 1), 45), ((5, 3, 2), 90), ((3, 1, 4), 90), ((4, 6, 2), 90))
    leg("150bp score", buf, meta, sets, False, k=2000)
    leg("150bp+cigar", buf, meta, sets, True, k=2000)
