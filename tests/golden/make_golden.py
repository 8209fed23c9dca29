#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference tree (run in the authoring container only).

Fixtures are DATA: the reference tests' own input files and expected outputs, and outputs of the
reference's WFA2-lib (compiled in place by oracle/Makefile into oracle/_ref) on those inputs and
on seeded synthetic pairs.  No reference source text is stored.

  wfa.utest.seq                    copy of tests/data/wfa.utest.seq (== external/WFA/tests/wfa.utest.seq)
  utest.affine.p{0,1,2}.alg        copies of external/WFA/tests/wfa.utest.check/test.affine.p{0,1,2}.alg
                                   (score + CIGAR; penalties (1,2,1), (3,1,4), (5,3,2))
  utest.score.affine.p{0,1,2}.alg  copies of tests/data/results/test.score.affine.p{0,1,2}.alg
  utest.affine.g231.alg            _ref output on wfa.utest.seq with the CLI default penalties (2,3,1)
  seq1k.seq / seq1k.*.scores       first 300 pairs and golden scores of tests/data/sequences_1000.h
  seq10k.seq / seq10k.*.scores     first 30 pairs and golden scores of tests/data/sequences_10K.h
  hifi.seq / hifi.g231.alg         first 12 pairs of tests/data/test_hifi.seq + _ref output
  synth.cfg{2,3}.alg               _ref output on seeded synthetic pairs (generator parameters inside)
  synth.cfg{4,5}.alg               the same for the first 8 / 4 pairs of the long-read sets (10 kbp @ 3 %, 30 kbp @ 10 %)
"""
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings"))
REF = os.environ.get("WFA_REFERENCE", "/root/reference")

import oracle_lib  # noqa: E402
import wfagpu  # noqa: E402


def write_alg(path, scores, cigars):
    with open(path, "w") as f:
        for s, c in zip(scores, cigars):
            f.write(f"{-int(s)}\t{c}\n")


def header_arrays(path):
    txt = open(path).read()
    seqs = re.findall(r'"([ACGTN]*)"', txt)
    arrays = {}
    for m in re.finditer(r"static const int (\w+)\[\d+\] = \{([^}]*)\}", txt):
        arrays[m.group(1)] = [int(v) for v in m.group(2).replace("\n", " ").split(",") if v.strip()]
    return seqs, arrays


def write_seq(path, pairs):
    with open(path, "wb") as f:
        for p, t in pairs:
            f.write(b">" + p + b"\n<" + t + b"\n")


def main():
    oracle_lib.build()
    assert oracle_lib.have_ref(), "oracle/_ref not built (reference tree missing?)"
    shutil.copyfile(f"{REF}/tests/data/wfa.utest.seq", f"{HERE}/wfa.utest.seq")
    for p in (0, 1, 2):
        shutil.copyfile(f"{REF}/external/WFA/tests/wfa.utest.check/test.affine.p{p}.alg", f"{HERE}/utest.affine.p{p}.alg")
        shutil.copyfile(f"{REF}/tests/data/results/test.score.affine.p{p}.alg", f"{HERE}/utest.score.affine.p{p}.alg")
    pairs = wfagpu.read_seq_file(f"{HERE}/wfa.utest.seq")
    buf, meta = wfagpu.layout_pairs(pairs)
    s, c = oracle_lib.ref_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    write_alg(f"{HERE}/utest.affine.g231.alg", s, c)

    # header fixtures -> .seq + scores (sequence order in the headers: pattern, text, pattern, ...)
    for name, hdr, keep in (("seq1k", "sequences_1000.h", 300), ("seq10k", "sequences_10K.h", 30)):
        seqs, arrays = header_arrays(f"{REF}/tests/data/{hdr}")
        prs = [(seqs[2 * i].encode(), seqs[2 * i + 1].encode()) for i in range(len(seqs) // 2)][:keep]
        write_seq(f"{HERE}/{name}.seq", prs)
        for arr_name, vals in arrays.items():
            pen = re.search(r"x(\d+)o(\d+)e(\d+)", arr_name).groups()
            np.savetxt(f"{HERE}/{name}.x{pen[0]}o{pen[1]}e{pen[2]}.scores", np.array(vals[:keep]), fmt="%d")

    hp = wfagpu.read_seq_file(f"{REF}/tests/data/test_hifi.seq", limit=12)
    write_seq(f"{HERE}/hifi.seq", hp)
    buf, meta = wfagpu.layout_pairs(hp)
    s, c = oracle_lib.ref_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    write_alg(f"{HERE}/hifi.g231.alg", s, c)

    # seeded synthetic sets shaped like BASELINE.json configs[1] and configs[2] (small N)
    for tag, n, length, err, seed in (("cfg2", 2000, 150, 0.02, 2), ("cfg3", 500, 1000, 0.05, 3)):
        buf, meta = wfagpu.generate_pairs(n, length, err, seed)
        s, c = oracle_lib.ref_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
        with open(f"{HERE}/synth.{tag}.alg", "w") as f:
            f.write(f"# generate_pairs(n={n}, length={length}, error={err}, seed={seed}) penalties=2,3,1\n")
            for sc, cg in zip(s, c):
                f.write(f"{-int(sc)}\t{cg}\n")
    # long reads (BASELINE.json configs[3]/[4] shapes): the first `keep` pairs of the set the GPU test regenerates
    for tag, n, keep, length, err, seed in (("cfg4", 64, 8, 10000, 0.03, 44), ("cfg5", 16, 4, 30000, 0.10, 55)):
        buf, meta = wfagpu.generate_pairs(n, length, err, seed)
        s, c = oracle_lib.ref_batch(buf, meta[:keep], (2, 3, 1), cigar=True, nthreads=4)
        with open(f"{HERE}/synth.{tag}.alg", "w") as f:
            f.write(f"# first {keep} pairs of generate_pairs(n={n}, length={length}, error={err}, seed={seed}) penalties=2,3,1\n")
            for sc, cg in zip(s, c):
                f.write(f"{-int(sc)}\t{cg}\n")
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
