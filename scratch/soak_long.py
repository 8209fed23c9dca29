"""Soak of the backtrace paths of long alignments against the oracle (scores and CIGARs): reads of 1.6-12 kbp, error rates to 15 %,
random penalties and budgets; tuning.trace_mode 0 (automatic), 2 (all in the wave-per-alignment kernel), 4 (walk there, replay by
lanes out of LDS); every set twice on the same context (buffers reused)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, wfagpu, oracle_lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = random.Random(seed)


def rand_pairs(n, lo, hi, err):
    out = []
    for _ in range(n):
        L = rng.randint(lo, hi)
        t = bytes(rng.choice(b"ACGT") for _ in range(L))
        p = bytearray(t)
        for _ in range(int(L * err) + rng.randint(0, 2)):
            op = rng.randint(0, 3)
            if op == 0 and p:
                p[rng.randrange(len(p))] = rng.choice(b"ACGT")
            elif op == 1 and p:
                a = rng.randrange(len(p)); del p[a:a + rng.randint(1, 8)]
            elif op == 2:
                a = rng.randint(0, len(p)); p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 8)))
            else:
                p.insert(rng.randint(0, len(p)), rng.choice(b"ACGT"))
        out.append((bytes(p), t) if rng.random() < 0.5 else (t, bytes(p)))
    return out


al = wfagpu.DeviceAligner(0)
bad = 0; split = 0
t0 = time.time()
for it in range(iters):
    pen = (rng.randint(1, 8), rng.randint(0, 10), rng.randint(1, 4))
    hi = rng.choice([2500, 4000, 8000, 12000])
    pairs = rand_pairs(rng.choice([40, 90, 130]), 1600, hi, rng.choice([0.0, 0.01, 0.05, 0.15]))
    pairs += rand_pairs(20, 0, 400, 0.1) + [(b"", b"ACGT" * 500), (b"ACGT" * 1000, b"ACGT" * 700), (b"A" * 2000, b"C" * 2000)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
    mode = rng.choice([0, 2, 4, 4])
    al.set_tuning(trace_mode=mode)
    batch = al.upload(buf, meta)
    for max_error in (rng.choice([50, 400, 3000]), 60000):
        for rep in range(2):
            s, c = al.align(batch, pen, max_error=max_error, compute_cigar=True)
            split += al.stats().pairs_trace_split
            if not np.array_equal(s, so) or c != co:
                bad += 1
                k = next(i for i in range(len(pairs)) if s[i] != so[i] or c[i] != co[i])
                print("MISMATCH it", it, "pen", pen, "max_error", max_error, "mode", mode, "pair", k, len(pairs[k][0]), len(pairs[k][1]), s[k], so[k], flush=True)
print("long-read soak seed", seed, "iterations", iters, "mismatching runs", bad, "alignments walked by wavefronts and replayed by lanes:", split, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
