import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
al = wfagpu.DeviceAligner(0)
for name, gen, me in (("iid", lambda: wfagpu.generate_pairs(16384, 10000, 0.03, seed=5, nthreads=16), 3000),
                      ("hard", lambda: wfagpu.generate_pairs_model(4096, 10000, seed=6, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02, long_min=30, long_max=150, cluster=0.3, nthreads=16), 6000)):
    buf, meta = gen(); batch = al.upload(buf, meta)
    for beta in (352, 512, 1024):
        al.align(batch, (2,3,1), max_error=me, compute_cigar=True, band=25, band_width=beta, fetch=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        al.align(batch, (2,3,1), max_error=me, compute_cigar=True, band=25, band_width=beta, fetch=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = al.stats()
        print(name, "beta", beta, "%.1f ms" % (dt*1e3), "align %.1f" % st.align_ms, "tiers", list(st.pairs_tier), "bpc", st.blocks_per_cu_tier0, flush=True)
