import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
al = wfagpu.DeviceAligner(0)
for n in (256, 512, 1024, 2048):
    buf, meta = wfagpu.generate_pairs(n, 30000, 0.10, seed=5, nthreads=16)
    batch = al.upload(buf, meta)
    for cig in (True, False):
        al.align(batch, (2, 3, 1), max_error=9000, compute_cigar=cig, fetch=False)
        t0 = time.perf_counter(); al.align(batch, (2, 3, 1), max_error=9000, compute_cigar=cig, fetch=False); t1 = time.perf_counter()
        st = al.stats()
        print("cfg5 n", n, "cigar" if cig else "score", "wall %.1f ms align %.1f trace %.1f" % ((t1 - t0) * 1e3, st.align_ms, st.trace_ms), "passes", st.sub_batches,
              "launches", st.align_launches, "Gcells/s %.1f" % (st.cells / st.align_ms / 1e6), "pairs/s %.0f" % (n / (t1 - t0)), "arenaGB %.1f" % (st.arena_units * 16 / 1e9), flush=True)
