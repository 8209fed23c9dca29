"""A batch of mostly short reads with a few long ones: length buckets keep the short reads' residency."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
al = wfagpu.DeviceAligner(0)
b1, m1 = wfagpu.generate_pairs(200000, 150, 0.02, seed=1, nthreads=16)
b2, m2 = wfagpu.generate_pairs(200, 20000, 0.03, seed=2, nthreads=16)
pairs = wfagpu.pairs_from_layout(b1, m1) + wfagpu.pairs_from_layout(b2, m2)
import random; random.Random(5).shuffle(pairs)
buf, meta = wfagpu.layout_pairs(pairs)
batch = al.upload(buf, meta)
for cig in (True, False):
    al.align(batch, (2, 3, 1), max_error=6000, compute_cigar=cig, fetch=False)
    t0 = time.perf_counter(); s, c = al.align(batch, (2, 3, 1), max_error=6000, compute_cigar=cig, fetch=False); t1 = time.perf_counter()
    st = al.stats()
    print("mixed 200k x 150bp + 200 x 20kbp", "cigar" if cig else "score", "wall %.1f ms align %.1f trace %.1f" % ((t1 - t0) * 1e3, st.align_ms, st.trace_ms),
          "launches", st.align_launches, "tiers", list(st.pairs_tier), "passes", st.sub_batches, flush=True)
sc, cg = al.align(batch, (2, 3, 1), max_error=6000, compute_cigar=True)
idx = list(range(0, len(pairs), 997)) + [i for i, p in enumerate(pairs) if len(p[0]) > 5000][:6]
sub = [pairs[i] for i in idx]
bs, ms = wfagpu.layout_pairs(sub)
so, co, _ = oracle_lib.oracle_batch(bs, ms, (2, 3, 1), cigar=True, nthreads=16)
print("parity on", len(idx), "pairs:", all(sc[i] == so[j] and cg[i] == co[j] for j, i in enumerate(idx)))
