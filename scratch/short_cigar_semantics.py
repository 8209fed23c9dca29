"""Would a CIGAR-producing tier 5 (one diagonal per lane, no per-row limits: every I/D value past a sequence end becomes NULL when
it is computed) give WFA2's CIGARs?  The oracle with that one change (a SECOND library, built here with
-DORACLE_EXPERIMENT_NULL_INVALID_GAPS into scratch/: the checker the tests load never has it) against the oracle as it
is, on random short pairs; prints the first pairs whose CIGAR (or score) changes."""
import ctypes as C, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib, wfagpu
EXP_SO = os.path.join(ROOT, "scratch", "liboracle_nullgaps.so")
subprocess.run(["gcc", "-O3", "-march=x86-64-v2", "-fPIC", "-fopenmp", "-shared", "-DORACLE_EXPERIMENT_NULL_INVALID_GAPS",
                os.path.join(ROOT, "oracle", "wfa_oracle.c"), os.path.join(ROOT, "oracle", "band_oracle.c"), "-o", EXP_SO], check=True)
exp = C.CDLL(EXP_SO)
exp.oracle_batch.argtypes = oracle_lib.oracle().oracle_batch.argtypes
exp.oracle_batch.restype = C.c_int64


def exp_batch(buf, meta, pen):
    n = len(meta)
    off = oracle_lib._offsets(meta)
    scores = np.zeros(n, dtype=np.int32)
    cells = C.c_int64(0)
    stride = oracle_lib._cigar_stride(meta)
    cbuf = np.zeros(n * stride, dtype=np.uint8)
    buf = np.ascontiguousarray(buf)
    exp.oracle_batch(buf.ctypes.data, off.ctypes.data, n, pen[0], pen[1], pen[2], scores.ctypes.data, cbuf.ctypes.data, stride, C.byref(cells), 1)
    return scores, oracle_lib._split(cbuf, n, stride)


rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
total = diff = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    pen = rng.choice([(2, 3, 1), (1, 2, 1), (4, 6, 1), (1, 0, 1), (3, 1, 1), (7, 2, 1)] if len(sys.argv) <= 3 else [(5, 3, 2), (3, 1, 4), (7, 2, 3), (1, 0, 2), (2, 0, 3), (3, 5, 2), (8, 4, 4), (1, 4, 2), (6, 1, 3)])
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
    s0, c0, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=1)
    s1, c1 = exp_batch(buf, meta, pen)
    bad = [i for i in range(len(pairs)) if c0[i] != c1[i] or s0[i] != s1[i]]
    total += len(pairs); diff += len(bad)
    for i in bad[:2]:
        print(f"pen {pen}: pattern {pairs[i][0].decode()} text {pairs[i][1].decode()}: WFA2 {s0[i]} {c0[i]}  | nulled gaps {s1[i]} {c1[i]}", flush=True)
print(f"{diff} of {total} pairs differ")
