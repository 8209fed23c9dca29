import sys, os
sys.path.insert(0, '/root/repo/wfa-gpu_amd/bindings'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, wfagpu, oracle_lib
buf, meta = wfagpu.generate_pairs(64, 2000, 0.05, seed=3)
for cig in (True, False):
    al = wfagpu.DeviceAligner(0, min_tier=4)
    try:
        b = al.upload(buf, meta)
        s, c = al.align(b, (2, 3, 1), max_error=3000, compute_cigar=cig)
        st = al.stats()
        so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=cig, nthreads=4)
        print('cigar', cig, 'ok', np.array_equal(s, so), list(st.pairs_tier))
    except Exception as e:
        print('cigar', cig, 'FAILED', e)
    al.close()
