import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
al = wfagpu.DeviceAligner(0)
for tag, n, L, err, e, cig, nchk in (("cfg4-10k-3%", 16384, 10000, 0.03, 3000, True, 8), ("cfg4-10k-3% score", 16384, 10000, 0.03, 3000, False, 8),
                                    ("cfg5-30k-10%", 256, 30000, 0.10, 9000, True, 2), ("cfg5-30k-10% score", 256, 30000, 0.10, 9000, False, 2),
                                    ("10k-3% e=1000", 4096, 10000, 0.03, 1000, True, 0)):
    buf, meta = wfagpu.generate_pairs(n, L, err, seed=5)
    batch = al.upload(buf, meta)
    t0 = time.perf_counter(); s, c = al.align(batch, (2, 3, 1), max_error=e, compute_cigar=cig, fetch=False); t1 = time.perf_counter()
    t0 = time.perf_counter(); s, c = al.align(batch, (2, 3, 1), max_error=e, compute_cigar=cig, fetch=False); t1 = time.perf_counter()
    st = al.stats()
    print(tag, "n", n, "wall %.1f ms" % ((t1 - t0) * 1e3), "align %.1f trace %.1f pack %.2f" % (st.align_ms, st.trace_ms, st.pack_ms),
          "launches", st.align_launches, "tiers", list(st.pairs_tier), "retried", st.pairs_retried, "passes", st.sub_batches,
          "cells %.3g" % st.cells, "Gcells/s %.1f" % (st.cells / st.align_ms / 1e6), "pairs/s %.0f" % (n / (t1 - t0)),
          "arenaGB %.1f" % (st.arena_units * 16 / 1e9), flush=True)
    if nchk:
        sc, cg = al.align(batch, (2, 3, 1), max_error=e, compute_cigar=cig)
        so, co, _ = oracle_lib.oracle_batch(buf, meta[:nchk], (2, 3, 1), cigar=cig, nthreads=8)
        print("   parity:", np.array_equal(sc[:nchk], so), (cg[:nchk] == co) if cig else "-", "scores", so[:4], flush=True)

for beta, lam in ((512, 25), (256, 25), (1024, 50)):
    buf, meta = wfagpu.generate_pairs(4096, 10000, 0.03, seed=5)
    batch = al.upload(buf, meta)
    al.align(batch, (2, 3, 1), max_error=3000, compute_cigar=True, band=lam, band_width=beta, fetch=False)
    t0 = time.perf_counter(); s, c = al.align(batch, (2, 3, 1), max_error=3000, compute_cigar=True, band=lam, band_width=beta, fetch=False); t1 = time.perf_counter()
    st = al.stats()
    sc = s.cpu().numpy()
    so, _, _ = oracle_lib.oracle_batch(buf, meta[:256], (2, 3, 1), cigar=False, nthreads=16)
    print("banded 10k-3%% beta=%d lam=%d wall %.1f ms align %.1f trace %.1f pairs/s %.0f banded %d retried %d recall(256) %.3f lds %d bpc %d" % (beta, lam, (t1-t0)*1e3, st.align_ms, st.trace_ms, 4096/(t1-t0), st.pairs_banded, st.pairs_retried, float((sc[:256]==so).mean()), st.lds_bytes_tier0, st.blocks_per_cu_tier0), flush=True)
