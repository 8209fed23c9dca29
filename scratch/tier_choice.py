"""One-wave vs four-wave tier against ring size: exact + CIGAR on pairs of several lengths (DeviceAligner(t0_min_blocks=1) keeps tier 0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
al = wfagpu.DeviceAligner(0)
for n, L, err, me in ((262144, 2000, 0.05, 600), (65536, 3000, 0.05, 900), (65536, 5000, 0.04, 1200), (16384, 10000, 0.03, 3000), (16384, 5000, 0.08, 2400)):
    buf, meta = wfagpu.generate_pairs(n, L, err, seed=9, nthreads=16); batch = al.upload(buf, meta)
    al.align(batch, (2,3,1), max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    al.align(batch, (2,3,1), max_error=me, compute_cigar=True, fetch=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = al.stats()
    print(f"n {n} L {L} err {err}: {dt*1e3:.1f} ms align {st.align_ms:.1f} tiers {list(st.pairs_tier)} lds {st.lds_bytes_tier0} bpc {st.blocks_per_cu_tier0} budget {st.auto_budget}", flush=True)
