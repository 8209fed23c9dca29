"""Would a CIGAR-producing tier 5 (one diagonal per lane, no per-row limits: every I/D value past a sequence end becomes NULL when
it is computed) give WFA2's CIGARs?  The oracle with that one change (oracle_set_null_invalid_gaps) against the oracle as it
is, on random short pairs; prints the first pairs whose CIGAR (or score) changes."""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib, wfagpu
o = oracle_lib.oracle()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
total = diff = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    pen = rng.choice([(2, 3, 1), (1, 2, 1), (4, 6, 1), (1, 0, 1), (3, 1, 1), (7, 2, 1)])
    pairs = []
    for _ in range(20000):
        L = rng.randint(1, 60)
        t = bytes(rng.choice(b"ACGT") for _ in range(L))
        p = bytearray(t)
        for _ in range(rng.randint(0, 6)):
            r = rng.random(); a = rng.randint(0, len(p))
            if r < 0.3 and a < len(p): p[a] = rng.choice(b"ACGT")
            elif r < 0.65: del p[a:a + rng.randint(1, 6)]
            else: p[a:a] = bytes(rng.choice(b"AC") for _ in range(rng.randint(1, 6)))
        pairs.append((bytes(p), t))
    buf, meta = wfagpu.layout_pairs(pairs)
    o.oracle_set_null_invalid_gaps(0)
    s0, c0, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=1)
    o.oracle_set_null_invalid_gaps(1)
    s1, c1, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=1)
    o.oracle_set_null_invalid_gaps(0)
    bad = [i for i in range(len(pairs)) if c0[i] != c1[i] or s0[i] != s1[i]]
    total += len(pairs); diff += len(bad)
    for i in bad[:2]:
        print(f"pen {pen}: pattern {pairs[i][0].decode()} text {pairs[i][1].decode()}: WFA2 {s0[i]} {c0[i]}  | nulled gaps {s1[i]} {c1[i]}", flush=True)
print(f"{diff} of {total} pairs differ")
