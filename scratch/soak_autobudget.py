"""Auto-tuned budgets (batches >= 8192 pairs) on heterogeneous batches: mixtures of lengths and error rates make the
sampled budget wrong for whole sub-populations; every score and CIGAR must still equal the oracle's."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = random.Random(seed)
al = wfagpu.DeviceAligner(0)
bad = 0
for it in range(iters):
    pen = (rng.randint(1, 6), rng.randint(0, 8), rng.randint(1, 4))
    pops = [(rng.choice([60, 150, 400]), rng.choice([0.0, 0.01, 0.05, 0.2, 0.6])) for _ in range(rng.randint(1, 4))]
    pairs = []
    n = rng.choice([8192, 12000, 20000])
    for i in range(n):
        L, err = pops[0] if rng.random() < 0.9 else rng.choice(pops)    # 90 % from one population: the sample misses the rest
        t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(L // 2, L)))
        p = bytearray(t)
        for _ in range(int(len(t) * err) + rng.randint(0, 1)):
            op = rng.randint(0, 2)
            if op == 0 and p: p[rng.randrange(len(p))] = rng.choice(b"ACGT")
            elif op == 1 and p: del p[rng.randrange(len(p))]
            else: p.insert(rng.randint(0, len(p)), rng.choice(b"ACGT"))
        pairs.append((bytes(p), t))
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
    batch = al.upload(buf, meta)
    for me in (rng.choice([30, 100]), 2000):
        s, c = al.align(batch, pen, max_error=me, compute_cigar=True)
        st = al.stats()
        ok = np.array_equal(s, so) and c == co
        s2, _ = al.align(batch, pen, max_error=me, compute_cigar=False)
        ok2 = np.array_equal(s2, so)
        print("it", it, "n", n, "pen", pen, "pops", pops, "max_error", me, "auto", st.auto_budget, "missed", st.pairs_budget_missed, "retried", st.pairs_retried, "ok", ok, ok2, flush=True)
        bad += (not ok) + (not ok2)
print("auto-budget soak seed", seed, "failures", bad)
sys.exit(1 if bad else 0)
