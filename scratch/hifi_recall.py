"""Adaptive band on the reference's real HiFi-shaped test pairs (tests/golden/hifi.seq = tests/data/test_hifi.seq of the
reference): recall (share of optimal scores) and score excess per beta, next to README.md:133-134 of the reference
(96.8-99.9 % for beta in {348, 512, 1024} on its HiFi data)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, wfagpu, oracle_lib
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
pairs = wfagpu.read_seq_file(os.path.join(root, "tests", "golden", "hifi.seq"))
buf, meta = wfagpu.layout_pairs(pairs)
gold, _ = oracle_lib.read_alg(os.path.join(root, "tests", "golden", "hifi.g231.alg"))
print("pairs", len(pairs), "lengths", int(meta["pattern_len"].min()), "-", int(meta["pattern_len"].max()), "optimal scores", gold.min(), "-", gold.max())
for lam in (25, 50):
    for beta in (348, 512, 1024):
        al = wfagpu.DeviceAligner(0, force_band=1)
        batch = al.upload(buf, meta)
        s, c = al.align(batch, (2, 3, 1), max_error=4000, compute_cigar=True, band=lam, band_width=beta)
        st = al.stats()
        ok = all(oracle_lib.check_cigar(p, t, cg, (2, 3, 1)) == (True, int(sc)) for (p, t), cg, sc in zip(pairs, c, s))
        ex = (s - gold)
        print(f"lambda {lam:3d} beta {beta:5d}: recall {float((s == gold).mean()):.3f} ({int((s == gold).sum())}/{len(gold)}), banded pairs {st.pairs_banded}, "
              f"mean excess {ex.mean():.1f}, max excess {ex.max()}, valid+cost==score {ok}", flush=True)
        al.close()
