"""Soak of the short-wavefront tier (tier 5): random short pairs, score-only AND with CIGARs, penalty sets the tier is compiled
for (e == 1 and max(x, o + e) <= 8 after the common-factor reduction), budgets small enough for 16- and 32-lane groups, batches
above and below the budget-tuning threshold, a tiny arena now and then (NOMEM passes); every score and every CIGAR against the checker.
  python scratch/soak_short.py <seed> <iterations>"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wfa-gpu_amd", "bindings"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, wfagpu, oracle_lib
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = random.Random(seed)
al = wfagpu.DeviceAligner(0)
bad = 0; used = 0; used_c = 0; t0 = time.time()
def rand_pair(maxlen, err):
    t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, maxlen)))
    p = bytearray(t)
    for _ in range(int(len(t) * err) + rng.randint(0, 2)):
        op = rng.randint(0, 2)
        if op == 0 and p: p[rng.randrange(len(p))] = rng.choice(b"ACGT")
        elif op == 1 and p: a = rng.randrange(len(p)); del p[a:a + rng.randint(1, 6)]
        else: a = rng.randint(0, len(p)); p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 6)))
    return (bytes(p), t) if rng.random() < 0.5 else (t, bytes(p))
for it in range(iters):
    pen = rng.choice([(2, 3, 1), (4, 6, 2), (1, 2, 1), (6, 9, 3), (3, 6, 3), (1, 0, 1), (4, 6, 1), (7, 7, 1), (8, 2, 1), (5, 1, 1), (3, 4, 1),
                      (5, 3, 2), (3, 1, 4), (7, 2, 3), (3, 5, 2), (1, 0, 2), (2, 0, 3), (8, 4, 4), (1, 4, 2), (6, 1, 3), (2, 2, 4), (5, 0, 3)])      # (round 6: gap extensions 2..4 on tier 5)
    n = rng.choice([64, 500, 9000, 12000])
    maxlen = rng.choice([30, 150, 300, 600])
    err = rng.choice([0.0, 0.01, 0.03, 0.08])
    base = [rand_pair(maxlen, err) for _ in range(min(n, 600))]
    pairs = [base[i % len(base)] for i in range(n)]
    pairs[0] = (b"", b""); pairs[1] = (b"A", b""); pairs[2] = (b"ACGT" * 10, b"ACGT" * 10)
    # (round 5) a few pairs with bytes outside ACGT (the byte-compare class: flagged by tier 5 itself where it packs its own reads), the
    # older paths of the one-kernel call now and then, budgets inherited from the previous call of the stream now and then
    for q in range(3, n, rng.choice([97, 911, 100000])):
        pq, tq = pairs[q]
        if pq: pq = bytearray(pq); pq[rng.randrange(len(pq))] = rng.choice(b"NnRYKM"); pairs[q] = (bytes(pq), tq)
    al.set_tuning(**rng.choice([{}, {}, {"no_host_parts": 1}, {"no_fused_pack": 1}, {"short_iterations": rng.choice([2, 7])}, {"no_short_cigar": 1}]))
    al.hint_same_stream(rng.random() < 0.5)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
    batch = al.upload(buf, meta)
    g = {(4, 6, 2): 2, (6, 9, 3): 3, (3, 6, 3): 3}.get(pen, 1) * max(1, max(pen[0], pen[1] + pen[2]) // 4)
    for me in (rng.choice([3, 8, 14]) * g, rng.choice([20, 30, 45]) * g, 2000):
        s, _ = al.align(batch, pen, max_error=me, compute_cigar=False)
        used += al.stats().pairs_tier[5]
        if not np.array_equal(s, so):
            bad += 1
            k = int(np.nonzero(s != so)[0][0])
            print("MISMATCH it", it, "pen", pen, "max_error", me, "n", n, "pair", k, pairs[k], int(s[k]), int(so[k]), flush=True)
        s2, c2 = al.align(batch, pen, max_error=me, compute_cigar=True)
        used_c += al.stats().pairs_tier[5]
        if not np.array_equal(s2, so) or c2 != co:
            bad += 1
            k = next(i for i in range(n) if s2[i] != so[i] or c2[i] != co[i])
            print("CIGAR MISMATCH it", it, "pen", pen, "max_error", me, "n", n, "pair", k, pairs[k], int(s2[k]), int(so[k]), c2[k], co[k], flush=True)
print("short-tier soak seed", seed, "iterations", iters, "mismatching runs", bad, "pairs finished in tier 5:", used, "with CIGARs:", used_c, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
