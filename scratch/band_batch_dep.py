import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, wfagpu
al = wfagpu.DeviceAligner(0)
buf, meta = wfagpu.generate_pairs_model(4096, 10000, seed=6, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02, long_min=30, long_max=150, cluster=0.3)
b4 = al.upload(buf, meta); b1 = al.upload(buf, meta[:1024])
ex4, _ = al.align(b4, (2,3,1), max_error=6000, compute_cigar=False)
for beta, lam in ((1024, 10), (512, 10)):
    s4, _ = al.align(b4, (2,3,1), max_error=6000, compute_cigar=True, band=lam, band_width=beta)
    s1, _ = al.align(b1, (2,3,1), max_error=6000, compute_cigar=True, band=lam, band_width=beta)
    s4b, _ = al.align(b4, (2,3,1), max_error=6000, compute_cigar=False, band=lam, band_width=beta)
    print(beta, lam, "recall all", (s4 == ex4).mean(), "first1024 of big", (s4[:1024] == ex4[:1024]).mean(), "small batch", (s1 == ex4[:1024]).mean(),
          "same as big?", np.array_equal(s1, s4[:1024]), "ndiff", int((s1 != s4[:1024]).sum()), "score-only recall", (s4b == ex4).mean(),
          "per-quarter recall", [(s4[i*1024:(i+1)*1024] == ex4[i*1024:(i+1)*1024]).mean() for i in range(4)])
