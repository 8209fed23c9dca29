import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
al = wfagpu.DeviceAligner(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
buf, meta = wfagpu.generate_pairs(n, 10000, 0.03, seed=5, nthreads=16)
batch = al.upload(buf, meta)
so, _, _ = oracle_lib.oracle_batch(buf, meta[:512], (2, 3, 1), cigar=False, nthreads=16)
for beta, lam in ((512, 25), (256, 25), (128, 25), (1024, 50)):
    for cig in (True, False):
        al.align(batch, (2, 3, 1), max_error=3000, compute_cigar=cig, band=lam, band_width=beta, fetch=False)
        t0 = time.perf_counter(); s, c = al.align(batch, (2, 3, 1), max_error=3000, compute_cigar=cig, band=lam, band_width=beta, fetch=False); t1 = time.perf_counter()
        st = al.stats()
        sc = s.cpu().numpy()
        print("banded 10k-3%% n %d beta=%d lam=%d %s wall %.1f ms align %.1f trace %.1f pairs/s %.0f banded %d retried %d recall(512) %.4f Gcells/s %.1f" % (
            n, beta, lam, "cigar" if cig else "score", (t1-t0)*1e3, st.align_ms, st.trace_ms, n/(t1-t0), st.pairs_banded, st.pairs_retried, float((sc[:512]==so).mean()), st.cells/st.align_ms/1e6), flush=True)
