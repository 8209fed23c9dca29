import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu
al = wfagpu.DeviceAligner(0)
n = 1000000
buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000)
batch = al.upload(buf, meta)
for cig in (False,):
    for e in (2000, 600, 300, 250, 200, 184):
        al.align(batch, (2, 3, 1), max_error=e, compute_cigar=cig, fetch=False)
        al.align(batch, (2, 3, 1), max_error=e, compute_cigar=cig, fetch=False)
        st = al.stats()
        print("cigar", cig, "e", e, "align %.2f ms total %.2f cells %.4g bpc %d lds %d" % (st.align_ms, st.total_ms, st.cells, st.blocks_per_cu_tier0, st.lds_bytes_tier0), "retried", st.pairs_retried, "launches", st.align_launches, flush=True)
