"""Randomised differential soak against the oracle (scores and CIGARs), wider than tests/test_gpu_parity.py's
stress test: more penalty triples, lengths, error rates, budgets; score-only and CIGAR; tier override."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = random.Random(seed)


def rand_pairs(n, maxlen, err):
    out = []
    for _ in range(n):
        L = rng.randint(0, maxlen)
        t = bytes(rng.choice(b"ACGT") for _ in range(L))
        p = bytearray(t)
        for _ in range(int(L * err) + rng.randint(0, 2)):
            op = rng.randint(0, 3)
            if op == 0 and p:
                p[rng.randrange(len(p))] = rng.choice(b"ACGT")
            elif op == 1 and p:
                a = rng.randrange(len(p)); del p[a:a + rng.randint(1, 6)]
            elif op == 2:
                a = rng.randint(0, len(p)); p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 6)))
            else:
                p.insert(rng.randint(0, len(p)), rng.choice(b"ACGT"))
        out.append((bytes(p), t) if rng.random() < 0.5 else (t, bytes(p)))
    return out


al = wfagpu.DeviceAligner(0)
bad = 0
t0 = time.time()
for it in range(iters):
    pen = (rng.randint(1, 12), rng.randint(0, 15), rng.randint(1, 8))
    maxlen = rng.choice([40, 150, 400, 900, 2500])
    pairs = rand_pairs(rng.choice([64, 200, 600]), maxlen, rng.choice([0.0, 0.02, 0.08, 0.2, 0.5]))
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT" * 5), (b"ACGT" * 30, b"TGCA" * 30)]
    for _ in range(12):   # very different lengths: |kend| large, in both directions
        a = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 60)))
        bb = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(100, maxlen + 100)))
        cut = rng.randint(0, len(bb))
        emb = bb[:cut] + a + bb[cut:]
        pairs.append(rng.choice([(a, bb), (bb, a), (bb, emb), (emb, bb)]))
    for _ in range(6):    # low-complexity
        u = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 4)))
        pairs.append((u * rng.randint(1, 80), u * rng.randint(1, 80)))
    if rng.random() < 0.3:
        pairs += [(bytes(rng.choice(b"ACGTN") for _ in range(rng.randint(1, 200))), bytes(rng.choice(b"ACGTNacgt") for _ in range(rng.randint(1, 200)))) for _ in range(20)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
    min_tier = rng.randint(1, 4) if rng.random() < 0.25 else 0
    # (the backtrace of long alignments: automatic, all of it in the wave-per-alignment kernel, or walk there + replay by lanes)
    al.set_tuning(min_tier=min_tier, trace_mode=rng.choice([0, 0, 2, 4, 4]))
    # (every other clean set arrives packed on the host: wfagpu_amd_batch_t::d_packed, no ASCII on the device)
    clean = all(set(p) <= set(b"ACGT") and set(t) <= set(b"ACGT") for p, t in pairs)
    batch = al.upload_packed(buf, meta) if clean and rng.random() < 0.5 else al.upload(buf, meta)
    for max_error in (rng.choice([1, 5, 20]), rng.choice([60, 200, 1000]), 20000):
        s, c = al.align(batch, pen, max_error=max_error, compute_cigar=True)
        if not np.array_equal(s, so) or c != co:
            bad += 1
            k = next(i for i in range(len(pairs)) if s[i] != so[i] or c[i] != co[i])
            print("MISMATCH it", it, "pen", pen, "max_error", max_error, "tier", min_tier, "pair", k, pairs[k], s[k], so[k], c[k], co[k], flush=True)
    s2, _ = al.align(batch, pen, max_error=rng.choice([3, 50, 3000]), compute_cigar=False)
    if not np.array_equal(s2, so):
        bad += 1
        print("SCORE MISMATCH it", it, "pen", pen, flush=True)
print("soak seed", seed, "iterations", iters, "mismatching runs", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
