"""GPU parity: every stage of the hot path, called through the C-ABI, against the oracle and the
committed golden vectors.  Bit-exact (integer scores, byte-identical CIGAR strings)."""
import os
import random

import numpy as np
import pytest

import oracle_lib
import wfagpu
from test_oracle import PENS, _rand_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aligner():
    al = wfagpu.DeviceAligner(0)
    yield al
    al.close()


def _run(aligner, buf, meta, pen, max_error, cigar=True):
    batch = aligner.upload(buf, meta)
    return aligner.align(batch, pen, max_error=max_error, compute_cigar=cigar)


@pytest.mark.parametrize("max_len", [60, 110, 240, 300, 500, 1500])
def test_pack_literals_and_random(aligner, max_len):
    """tests/test_packing_kernel.cu of the reference: literal pairs + bit layout; here every 2-bit field is
    compared with the oracle's packer (little-endian word layout of this build).  The longest sequence of the batch picks
    the lanes a pair gets in the pack kernel (8, 16, 32 or 64: 8, 4, 2 or 1 pairs per wavefront): every width, with pair counts
    that leave the last wavefront partly empty."""
    import ctypes as C
    rng = random.Random(5 + max_len)
    pairs = [(b"GATTACA", b"GATACA"), (b"ACGT" * 9, b"ACGT" * 9 + b"A"), (b"T" * 33, b"C" * 16), (b"A", b"G")]
    pairs += _rand_pairs(rng, 201, max_len) + [(b"ACGNACGT", b"acgt"), (b"", b"A"), (b"", b""), (b"C" * (max_len - 1), b"")]
    buf, meta = wfagpu.layout_pairs(pairs)
    batch = aligner.upload(buf, meta)
    packed, flags = aligner.pack(batch)
    hm = batch._meta_host
    o = oracle_lib.oracle()
    for i, (p, t) in enumerate(pairs):
        for which, (seq, off) in enumerate(((p, int(hm[i]["pattern_offset_packed"])), (t, int(hm[i]["text_offset_packed"])))):
            nw = (len(seq) + 15) // 16
            words = (C.c_uint32 * (nw + 1))()
            bad = o.oracle_pack2(seq, len(seq), words)
            got = packed[off // 4: off // 4 + nw + 1]
            assert flags[2 * i + which] == bad, (i, which)
            if not bad:
                assert list(got[:nw]) == list(words)[:nw], (i, which)
            assert got[nw] == 0  # spare word


def test_host_packed_batches(aligner, golden_dir):
    """wfagpu_amd_batch_t::d_packed: the words utils/host_pack.c writes are the pack kernel's, bit for bit, and a batch
    that arrives packed (no ASCII on the device at all) aligns to the oracle's scores and CIGARs."""
    rng = random.Random(17)
    pairs = [(b"GATTACA", b"GATACA"), (b"ACGT" * 9, b"ACGT" * 9 + b"A"), (b"T" * 33, b"C" * 16), (b"A", b"G"), (b"", b"ACG"), (b"C" * 16, b"")]
    pairs += _rand_pairs(rng, 300, 400) + wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))[:120]
    buf, meta = wfagpu.layout_pairs(pairs)
    batch = aligner.upload(buf, meta)
    dev_words, dev_flags = aligner.pack(batch)
    for scalar in (False, True):
        words, flags = wfagpu.host_pack(buf, batch._meta_host, batch.packed_bytes, scalar=scalar)
        assert not flags.any() and not dev_flags.any()
        assert np.array_equal(words[:batch.packed_bytes // 4], dev_words[:batch.packed_bytes // 4])
    pb = aligner.upload_packed(buf, meta)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for cigar in (True, False):
        s, c = aligner.align(pb, (2, 3, 1), max_error=300, compute_cigar=cigar)
        assert np.array_equal(s, np.asarray(so))
        if cigar:
            assert c == co
    with pytest.raises(ValueError):
        aligner.upload_packed(*wfagpu.layout_pairs([(b"ACGN", b"ACGT")]))


@pytest.mark.parametrize("tag", ["p0", "p1", "p2", "g231"])
def test_utest_goldens_cigar(aligner, golden_dir, tag):
    """WFA2's own unit-test goldens (score + CIGAR) for the reference's wfa.utest.seq; -e small on purpose so
    that the long pairs go through every escalation tier (the reference sends them to the CPU, tests/test-aligner.sh:27)."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "wfa.utest.seq"))
    buf, meta = wfagpu.layout_pairs(pairs)
    gs, gc = oracle_lib.read_alg(os.path.join(golden_dir, f"utest.affine.{tag}.alg"))
    s, c = _run(aligner, buf, meta, PENS[tag], max_error=25)
    assert np.array_equal(s, gs)
    assert c == gc


@pytest.mark.parametrize("tag", ["p0", "p1", "p2"])
def test_utest_goldens_score_only(aligner, golden_dir, tag):
    """tests/test-aligner.sh of the reference: -g 1,2,1 / 3,1,4 / 5,3,2 with -e 10000, score-only."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "wfa.utest.seq"))
    buf, meta = wfagpu.layout_pairs(pairs)
    gs, _ = oracle_lib.read_alg(os.path.join(golden_dir, f"utest.score.affine.{tag}.alg"))
    s, _ = _run(aligner, buf, meta, PENS[tag], max_error=10000, cigar=False)
    assert np.array_equal(s, gs)


@pytest.mark.parametrize("name,pens", [("seq1k", [(2, 3, 1), (5, 3, 2)]), ("seq10k", [(2, 3, 1), (3, 5, 2)])])
def test_api_goldens(aligner, golden_dir, name, pens):
    """tests/test_api.c golden score arrays, CIGAR and distance-only modes."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, f"{name}.seq"))
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen in pens:
        gold = -np.loadtxt(os.path.join(golden_dir, f"{name}.x{pen[0]}o{pen[1]}e{pen[2]}.scores"), dtype=np.int64)
        for cigar in (False, True):
            max_len = 1000 if name == "seq1k" else 10000
            s, c = _run(aligner, buf, meta, pen, max_error=int(0.1 * max_len * max(pen)), cigar=cigar)
            assert np.array_equal(s, gold[:len(s)])
            if cigar:
                for (p, t), cg, sc in zip(pairs, c, s):
                    ok, cost = oracle_lib.check_cigar(p, t, cg, pen)
                    assert ok and cost == sc


@pytest.mark.parametrize("fname,n,length,err,seed", [("synth.cfg2.alg", 2000, 150, 0.02, 2), ("synth.cfg3.alg", 500, 1000, 0.05, 3)])
def test_synthetic_config_goldens(aligner, golden_dir, fname, n, length, err, seed):
    """Seeded synthetic sets shaped like BASELINE.json configs[1]/[2]; expected output from the compiled reference."""
    buf, meta = wfagpu.generate_pairs(n, length, err, seed)
    gs, gc = oracle_lib.read_alg_skip_comments(os.path.join(golden_dir, fname))
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=max(50, int(0.1 * length * 3)))
    assert np.array_equal(s, gs)
    assert c == gc


def test_hifi_goldens(aligner, golden_dir):
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))
    buf, meta = wfagpu.layout_pairs(pairs)
    gs, gc = oracle_lib.read_alg(os.path.join(golden_dir, "hifi.g231.alg"))
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=3000)
    assert np.array_equal(s, gs)
    assert c == gc


@pytest.mark.parametrize("pen", [(2, 3, 1), (1, 2, 1), (3, 1, 4), (5, 3, 2), (4, 6, 2), (1, 0, 1), (7, 2, 3)])
def test_random_vs_oracle(aligner, pen):
    """Ragged random pairs incl. empty and 1-base sequences and unrelated pairs."""
    rng = random.Random(99 + sum(pen))
    pairs = _rand_pairs(rng, 400, 80) + _rand_pairs(rng, 80, 500, err=0.25) + _rand_pairs(rng, 10, 3000, err=0.15)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"A", b"A"), (b"A", b"C"), (b"ACGT", b"TGCA"),
              (b"AAAAAAAAAA", b"TTTTTTTTTTTTTTT"), (b"ACGTACGTAC", b"ACGTACGTACGTACGTACGT")]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, cells = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
    for max_error in (8, 200):
        s, c = _run(aligner, buf, meta, pen, max_error=max_error)
        assert np.array_equal(s, so)
        assert c == co
        s2, _ = _run(aligner, buf, meta, pen, max_error=max_error, cigar=False)
        assert np.array_equal(s2, so)


@pytest.mark.parametrize("pen", [(2, 3, 1), (5, 3, 2)])
def test_non_acgt_pairs_use_byte_compare_kernels(aligner, pen):
    """Pairs with N / lower case / IUPAC bytes: WFA2 compares raw bytes (N==N matches, 'a' != 'A'); the reference
    sends such pairs to the CPU (sequence_alignment_kernel.cu:474-482), here they run in the byte-compare kernels.
    Mixed with ACGT-only pairs in one batch; also long enough to need tier escalation."""
    rng = random.Random(4242)
    pairs = _rand_pairs(rng, 150, 120, alphabet=b"ACGTN") + _rand_pairs(rng, 60, 200, alphabet=b"ACGTacgtNRY")
    pairs += _rand_pairs(rng, 100, 150) + _rand_pairs(rng, 8, 2500, alphabet=b"ACGTN", err=0.12)
    pairs += [(b"N", b"N"), (b"NNNN", b"NNAN"), (b"acgt", b"ACGT"), (b"ACGTN", b""), (b"", b"N")]
    rng.shuffle(pairs)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
    for max_error in (10, 300):
        s, c = _run(aligner, buf, meta, pen, max_error=max_error)
        assert np.array_equal(s, so)
        assert c == co
        assert aligner.stats().pairs_raw > 200
        s2, _ = _run(aligner, buf, meta, pen, max_error=max_error, cigar=False)
        assert np.array_equal(s2, so)


@pytest.mark.parametrize("min_tier,pen", [(0, (2, 3, 1)), (1, (2, 3, 1)), (4, (2, 3, 1)), (0, (5, 3, 2)), (2, (5, 3, 2))])
def test_small_arena_forces_multiple_passes(golden_dir, min_tier, pen):
    """Backtrace arena smaller than the batch needs: pairs are re-queued for a further pass, results identical -- in the
    one-wave, the multi-wave and the hybrid tier, for gap extension 1 and beyond (each has its own lean loop, and each of
    those claims its arena row before it commits a score)."""
    al = wfagpu.DeviceAligner(0, arena_bytes=8 << 20, min_tier=min_tier)
    try:
        buf, meta = wfagpu.generate_pairs(3000 if min_tier == 0 else 1200, 1000, 0.05, 11)
        so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
        batch = al.upload(buf, meta)
        s, c = al.align(batch, pen, max_error=300 * max(1, pen[0] // 2), compute_cigar=True)
        assert al.stats().sub_batches > 1
        assert np.array_equal(s, so)
        assert c == co
    finally:
        al.close()


def test_full_size_properties(aligner):
    """BASELINE configs[1] at full size (100k x 150 bp): size-independent properties -- every CIGAR replays onto
    its pair, its gap-affine cost equals the reported score, and a 2000-pair sample equals the oracle."""
    n = 100000
    buf, meta = wfagpu.generate_pairs(n, 150, 0.02, 21)
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=45)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    idx = np.random.RandomState(0).choice(n, 2000, replace=False)
    sub = [pairs[i] for i in idx]
    sb, sm = wfagpu.layout_pairs(sub)
    so, co, _ = oracle_lib.oracle_batch(sb, sm, (2, 3, 1), cigar=True, nthreads=8)
    assert np.array_equal(s[idx], so)
    assert [c[i] for i in idx] == co
    for i in range(0, n, 7):
        ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], c[i], (2, 3, 1))
        assert ok and cost == s[i]


@pytest.mark.parametrize("n,length,err,beta,lam,max_error,min_recall", [
    # recall floors: 100 % on i.i.d. single-base edits at every beta/lambda, for the reference's rule and for this build alike
    # (profiles/r04/banded.md), less 1 %
    (512, 10000, 0.03, 512, 25, 3000, 0.99),     # BASELINE configs[3] shape: -e 3000 -t 512 -B auto
    (2000, 1000, 0.05, 128, 25, 300, 0.99),
    (2000, 1000, 0.05, 64, 10, 300, 0.99),
    (300, 3000, 0.15, 256, 50, 2000, 0.99),
])
def test_adaptive_band_is_valid_and_mostly_optimal(aligner, n, length, err, beta, lam, max_error, min_recall):
    """the reference's banded mode has no bit-defined output, so parity = every CIGAR replays onto its
    pair, its gap-affine cost equals the reported score, the score is never below the optimum, and recall (share of
    optimal scores) is in the reference's range (img/approximate-recall.png: 96.8-99.9 % on its data).  Also: the result
    is deterministic, and pairs the band cannot finish are finished exactly on the GPU."""
    buf, meta = wfagpu.generate_pairs(n, length, err, seed=31)
    batch = aligner.upload(buf, meta)
    s, c = aligner.align(batch, (2, 3, 1), max_error=max_error, compute_cigar=True, band=lam, band_width=beta)
    st = aligner.stats()
    assert st.pairs_banded > 0
    s2, c2 = aligner.align(batch, (2, 3, 1), max_error=max_error, compute_cigar=True, band=lam, band_width=beta)
    assert np.array_equal(s, s2) and c == c2
    so, _, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=False, nthreads=8)
    assert (s >= so).all()
    recall = float((s == so).mean())
    print(f"banded beta={beta} lambda={lam} L={length} err={err}: recall {recall:.4f}, banded pairs {st.pairs_banded}/{n}")
    assert recall >= min_recall
    for (p, t), cg, sc in zip(wfagpu.pairs_from_layout(buf, meta), c, s):
        ok, cost = oracle_lib.check_cigar(p, t, cg, (2, 3, 1))
        assert ok and cost == sc
    s3, _ = aligner.align(batch, (2, 3, 1), max_error=max_error, compute_cigar=False, band=lam, band_width=beta)
    assert np.array_equal(s3, s)


@pytest.mark.parametrize("pen", [(2, 3, 1), (1, 0, 1), (5, 3, 2), (3, 1, 4)])
def test_score_budget_window_is_exact(aligner, pen):
    """The exact kernels only keep the diagonals an alignment of score <= max_error can visit.  Run every pair with
    max_error EQUAL to its optimal score (the tightest window that must still succeed, no escalation) and with
    max_error one below (must escalate): score and CIGAR stay byte-identical to WFA2's."""
    rng = random.Random(2024 + sum(pen))
    pairs = _rand_pairs(rng, 500, 120, err=0.2) + _rand_pairs(rng, 100, 600, err=0.1)
    # long gaps: the window is asymmetric around kend/2
    pairs += [(bytes(rng.choice(b"ACGT") for _ in range(60)), bytes(rng.choice(b"ACGT") for _ in range(60 + d))) for d in range(0, 40, 3)]
    pairs += [(bytes(rng.choice(b"ACGT") for _ in range(60 + d)), bytes(rng.choice(b"ACGT") for _ in range(60))) for d in range(0, 40, 3)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
    groups = {}
    for i in np.argsort(so, kind="stable"):
        groups.setdefault(int(so[i]), []).append(int(i))
    checked = 0
    for score, idx in sorted(groups.items()):
        if score == 0 or checked > 60:
            continue
        sub = [pairs[i] for i in idx]
        sb, sm = wfagpu.layout_pairs(sub)
        for me in (score, max(1, score - 1)):
            s, c = _run(aligner, sb, sm, pen, max_error=me)
            assert np.array_equal(s, so[idx]) and c == [co[i] for i in idx], (score, me)
            if me == score:
                assert aligner.stats().pairs_retried == 0, (score, me)
            else:
                assert aligner.stats().pairs_retried == len(idx), (score, me)
        checked += 1


@pytest.mark.parametrize("pen", [(40, 2, 1), (64, 0, 1), (33, 30, 1)])
def test_wide_ring_penalties_keep_the_guard_cells(aligner, pen):
    """A mismatch penalty x > 32 gives a ring of more than 32 M rows, i.e. more than 64 guard cells per score (dm on
    each side): the one-wavefront lean loop clears them in two passes, and only while the budget's reach shrinks the
    wavefront -- so the budgets here are tight (each pair's own score and a little more).  Byte-identical to WFA2."""
    rng = random.Random(77 + pen[0])
    pairs = _rand_pairs(rng, 300, 200, err=0.08) + _rand_pairs(rng, 100, 700, err=0.05)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
    for me in (int(np.percentile(so, 50)) + 1, int(so.max()) + 3, 5000):
        s, c = _run(aligner, buf, meta, pen, max_error=me)
        assert np.array_equal(s, so) and c == co, me


def test_randomised_penalties_and_shapes(aligner):
    """Stress: 24 random (x,o,e) triples x ragged pair sets (similar, unrelated, one-sided gaps, very different
    lengths, homopolymers) x three max_error settings -- byte-identical scores and CIGARs vs the oracle.  Exercises the
    exact-trimming slow path (values past a sequence end), null scores (all-even penalties), the score-budget window
    with large |kend| and every escalation tier."""
    rng = random.Random(77)
    for it in range(24):
        pen = (rng.randint(1, 9), rng.randint(0, 12), rng.randint(1, 6))
        pairs = _rand_pairs(rng, 120, 150, err=rng.choice([0.02, 0.1, 0.3]))
        pairs += [(bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 90))),
                   bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 90)))) for _ in range(40)]        # unrelated
        for _ in range(20):                                                                                      # one long gap
            base = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(40, 200)))
            cut = rng.randint(0, len(base) // 2)
            gap = rng.randint(1, len(base) // 2)
            other = base[:cut] + base[cut + gap:]
            pairs.append((base, other) if rng.random() < 0.5 else (other, base))
        pairs += [(b"A" * rng.randint(1, 120), b"A" * rng.randint(1, 120)) for _ in range(10)]                  # homopolymers
        pairs += [(b"AC" * rng.randint(1, 60), b"CA" * rng.randint(1, 60)) for _ in range(6)]
        buf, meta = wfagpu.layout_pairs(pairs)
        so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
        for max_error in (1, 40, 5000):
            s, c = _run(aligner, buf, meta, pen, max_error=max_error)
            assert np.array_equal(s, so), (pen, max_error)
            assert c == co, (pen, max_error)
        s2, _ = _run(aligner, buf, meta, pen, max_error=40, cigar=False)
        assert np.array_equal(s2, so), pen


@pytest.mark.parametrize("min_tier", [1, 2, 3, 4])
def test_every_tier_gives_the_same_answers(min_tier):
    """The 4-wave, 16-wave, HBM-ring (16-bit offsets) and hybrid-ring (tier 4: M and I rings in LDS, D ring in HBM)
    instantiations on a ragged set that the one-wave tier normally takes: wfagpu_amd_tuning_t::min_tier makes the planner
    skip the smaller tiers."""
    rng = random.Random(1234 + min_tier)
    pairs = _rand_pairs(rng, 96, 400, err=0.08)
    pairs += [(b"ACGT" * 50, b"ACGT" * 20), (b"", b"ACGTAC"), (b"GATTACA", b""), (b"A" * 300, b"A" * 299 + b"C")]
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen in ((2, 3, 1), (4, 6, 2), (5, 3, 2)):
        so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
        al = wfagpu.DeviceAligner(0, min_tier=min_tier)
        try:
            s, c = _run(al, buf, meta, pen, max_error=600)
            st = al.stats()
            assert sum(st.pairs_tier[t] for t in range(min_tier)) == 0, list(st.pairs_tier)
            assert np.array_equal(s, so) and c == co, (min_tier, pen)
            s2, _ = _run(al, buf, meta, pen, max_error=600, cigar=False)
            assert np.array_equal(s2, so)
        finally:
            al.close()


def test_sequences_beyond_16_bit_offsets():
    """A pair longer than 32766 bases cannot use 16-bit offsets: the HBM-ring tier with 32-bit offsets takes it."""
    rng = random.Random(99)
    base = bytes(rng.choice(b"ACGT") for _ in range(33500))
    other = bytearray(base)
    for pos in sorted(rng.sample(range(100, 33000), 12), reverse=True):
        r = rng.random()
        if r < 0.4:
            other[pos] = ord("A") if other[pos] != ord("A") else ord("C")
        elif r < 0.7:
            del other[pos]
        else:
            other.insert(pos, ord("G"))
    pairs = [(base, bytes(other)), (base[:2000], bytes(other[:2100]))]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=2)
    al = wfagpu.DeviceAligner(0)
    try:
        s, c = _run(al, buf, meta, (2, 3, 1), max_error=200)
        assert list(al.stats().pairs_tier)[3] >= 1
        assert np.array_equal(s, so) and c == co
    finally:
        al.close()


def test_zero_and_negative_budgets_still_finish(aligner):
    """max_error is only a sizing hint: 0 or a negative value must not stall the escalation."""
    rng = random.Random(3)
    pairs = _rand_pairs(rng, 64, 200, err=0.1) + [(b"", b""), (b"ACGT", b"ACGT")]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for me in (0, -5):
        s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=me)
        assert np.array_equal(s, so) and c == co, me


@pytest.mark.parametrize("pen,beta,lam,err", [((4, 1, 1), 128, 10, 0.15), ((2, 4, 5), 128, 750, 0.15), ((7, 10, 1), 256, 25, 0.15),
                                               ((4, 2, 5), 64, 1, 0.05)])
def test_narrow_band_scores_follow_the_returned_alignment(aligner, pen, beta, lam, err):
    """A narrow adaptive band on divergent pairs with multi-base indels: the band can drop the gap-extension cell, so
    the wavefront path opens two gaps back to back; printed they are one gap.  The reported score must be the
    gap-affine cost of the CIGAR that is returned (the reference's -c rule, utils/verification.c:91-146)."""
    rng = random.Random(hash((pen, beta)) & 0xFFFF)
    pairs = []
    for _ in range(96):
        t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(2000, 4000)))
        p = bytearray(t)
        for _ in range(int(len(t) * err)):
            op = rng.randint(0, 2)
            if op == 0:
                p[rng.randrange(len(p))] = rng.choice(b"ACGT")
            elif op == 1:
                a = rng.randrange(len(p)); del p[a:a + rng.randint(1, 12)]
            else:
                a = rng.randint(0, len(p)); p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 12)))
        pairs.append((bytes(p), t))
    buf, meta = wfagpu.layout_pairs(pairs)
    so, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False, nthreads=8)
    batch = aligner.upload(buf, meta)
    me = int(4000 * 0.3 * max(pen))
    s1, c1 = aligner.align(batch, pen, max_error=me, compute_cigar=True, band=lam, band_width=beta)
    s2, c2 = aligner.align(batch, pen, max_error=me, compute_cigar=True, band=lam, band_width=beta)
    assert np.array_equal(s1, s2) and c1 == c2
    for (p, t), cg, sc, opt in zip(pairs, c1, s1, so):
        ok, cost = oracle_lib.check_cigar(p, t, cg, pen)
        assert ok and cost == sc and sc >= opt, (len(p), len(t), sc, opt, cost)


def test_length_buckets_mixed_batch(aligner):
    """One call with pairs from 0 to ~9 kbp: the host driver runs them in length buckets (<= 1024, <= 4096, rest);
    every pair must come back, in input order, with the oracle's score and CIGAR -- with and without non-ACGT pairs."""
    rng = random.Random(2024)
    pairs = _rand_pairs(rng, 300, 200, err=0.05) + _rand_pairs(rng, 30, 3000, err=0.05) + [(b"", b""), (b"", b"ACGT"), (b"T", b"")]
    for _ in range(6):
        t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(6000, 9000)))
        p = bytearray(t)
        for _ in range(len(t) // 25):
            a = rng.randrange(len(p))
            r = rng.random()
            if r < 0.4:
                p[a] = rng.choice(b"ACGT")
            elif r < 0.7:
                del p[a:a + rng.randint(1, 3)]
            else:
                p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 3)))
        pairs.append((bytes(p), t))
    pairs += [(b"ACGTNNACGT" * 30, b"ACGTNACGT" * 33), (b"acgt" * 1500, b"acgt" * 1490 + b"ACGT")]
    rng.shuffle(pairs)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for max_error in (50, 2000):
        s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=max_error)
        assert np.array_equal(s, so) and c == co, max_error
        s2, _ = _run(aligner, buf, meta, (2, 3, 1), max_error=max_error, cigar=False)
        assert np.array_equal(s2, so)
    assert aligner.stats().sub_batches >= 3


# ---------------------------------------------------------------------------------------------------------
# Round 2: the host orchestration the headline number runs through, and the long-read configurations.

def _truth(buf, meta, pen, nthreads=8):
    """Ground truth for long reads (score + CIGAR): the reference's own WFA2 compiled in place (oracle/_ref, travels as
    a prebuilt .so) when it is there -- it is ~5x faster than the restatement at 30 kbp -- else the oracle."""
    if oracle_lib.have_ref():
        return oracle_lib.ref_batch(buf, meta, pen, cigar=True, memory_mode=0, nthreads=nthreads)
    s, c, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=min(nthreads, 4))
    return s, c


def _concat_layouts(parts):
    """[(buf, meta)] -> one batch (records rebased), order preserved."""
    bufs, metas, base = [], [], 0
    for buf, meta in parts:
        m = meta.copy()
        m["pattern_offset"] += base
        m["text_offset"] += base
        pad = (-len(buf)) % 4
        bufs.append(buf)
        if pad:
            bufs.append(np.zeros(pad, dtype=np.uint8))
        base += len(buf) + pad
        metas.append(m)
    return np.concatenate(bufs + [np.zeros(16, dtype=np.uint8)]), np.concatenate(metas)


def test_auto_budget_path_on_a_heterogeneous_batch(aligner):
    """The sampled auto-budget driver (csrc/wfa_host.hip: k_sample -> sampled run -> k_ratio -> percentile -> k_budget
    -> budget round -> misses re-run with the caller's budget) only engages at >= 8192 pairs with a window > 128
    diagonals -- the regime bench.py's cfg3 runs in.  A 0.7 % sub-population with four times the error rate sits above
    the sampled 99th percentile, so its pairs MUST miss their budget and be re-run; lengths vary 3x inside the bucket.
    Scores and CIGARs byte-identical to the oracle on every pair."""
    parts = [wfagpu.generate_pairs(9000, 1000, 0.03, seed=101), wfagpu.generate_pairs(7000, 600, 0.04, seed=102),
             wfagpu.generate_pairs(110, 1000, 0.13, seed=103), wfagpu.generate_pairs(300, 350, 0.02, seed=104)]
    buf, meta = _concat_layouts(parts)
    perm = np.random.RandomState(5).permutation(len(meta))
    meta = np.ascontiguousarray(meta[perm])
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=16)
    batch = aligner.upload(buf, meta)
    s, c = aligner.align(batch, (2, 3, 1), max_error=400, compute_cigar=True)
    st = aligner.stats()
    assert st.auto_budget > 0, "the auto-budget path did not engage"
    assert st.pairs_budget_missed >= 100, st.pairs_budget_missed
    assert st.auto_budget < 400
    assert np.array_equal(s, so)
    assert c == co
    s2, _ = aligner.align(batch, (2, 3, 1), max_error=400, compute_cigar=False)
    assert aligner.stats().auto_budget > 0
    assert np.array_equal(s2, so)


def test_auto_budget_sample_with_a_tiny_arena():
    """with an arena too small for the SAMPLE of the auto-budget step (several passes, NOMEM re-queues)
    the sampled run used to write its leftovers over the bucket's own pending list; pairs were then never aligned and
    the call still returned 0.  >= 8192 pairs, CIGARs, 2 MiB arena, compared against the oracle pair by pair."""
    al = wfagpu.DeviceAligner(0, arena_bytes=2 << 20)
    try:
        buf, meta = wfagpu.generate_pairs(10240, 400, 0.05, seed=77)
        so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=16)
        batch = al.upload(buf, meta)
        s, c = al.align(batch, (2, 3, 1), max_error=300, compute_cigar=True)
        st = al.stats()
        assert st.auto_budget > 0 and st.sub_batches > 4, (st.auto_budget, st.sub_batches)
        assert np.array_equal(s, so)
        assert c == co
    finally:
        al.close()


def test_cfg5_ont_30kbp_exact_with_cigars(aligner, golden_dir):
    """BASELINE configs[4] shape: ONT-like 30 kbp pairs at 10 % error, exact (unbanded) + CIGAR, -e 9000
    (lib/wfa_types.h:59-64 and lib/sequence_alignment.cu:31-57 size the reference's backtrace for this case).  The first
    four pairs are pinned by a committed golden made with the reference's WFA2 (tests/golden/make_golden.py); all 16
    are compared with the checker available on the box."""
    buf, meta = wfagpu.generate_pairs(16, 30000, 0.10, seed=55)
    gs, gc = oracle_lib.read_alg_skip_comments(os.path.join(golden_dir, "synth.cfg5.alg"))
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=9000)
    st = aligner.stats()
    assert np.array_equal(s[:len(gs)], gs)
    assert c[:len(gc)] == gc
    so, co = _truth(buf, meta, (2, 3, 1))
    assert np.array_equal(s, so)
    assert c == co
    assert st.cells > 16 * 30_000_000          # the exact search, not a band
    s2, _ = _run(aligner, buf, meta, (2, 3, 1), max_error=9000, cigar=False)
    assert np.array_equal(s2, so)
    # -e too small by 20x: escalation on the device, same answers
    s3, c3 = _run(aligner, buf[:int(meta[3]["text_offset"]) + 30016], meta[:4], (2, 3, 1), max_error=450)
    assert np.array_equal(s3, so[:4]) and c3 == co[:4]


def test_cfg4_hifi_10kbp_exact_with_cigars(aligner, golden_dir):
    """BASELINE configs[3] shape without the band: 10 kbp pairs at 3 % error, exact + CIGAR, -e 3000."""
    buf, meta = wfagpu.generate_pairs(64, 10000, 0.03, seed=44)
    gs, gc = oracle_lib.read_alg_skip_comments(os.path.join(golden_dir, "synth.cfg4.alg"))
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=3000)
    assert np.array_equal(s[:len(gs)], gs)
    assert c[:len(gc)] == gc
    so, co = _truth(buf, meta, (2, 3, 1))
    assert np.array_equal(s, so)
    assert c == co


@pytest.mark.parametrize("pen,beta,lam", [((2, 3, 1), 1024, 10), ((2, 3, 1), 512, 25), ((2, 3, 1), 352, 100), ((1, 2, 1), 512, 10),
                                          ((3, 4, 1), 352, 25), ((2, 3, 1), 512, 750)])
def test_adaptive_band_matches_the_reference_rule_on_long_read_shaped_pairs(aligner, pen, beta, lam):
    """The banded kernels against the reference's adaptive-band kernel restated on the CPU (oracle/band_oracle.c:
    lib/kernels/sequence_distance_kernel_aband.cu), on data that defeats a band that does not follow the alignment: 10 kbp
    pairs with multi-base indels, a few long ones and clustered errors (profiles/r04/banded.md).  Score-only: a pair the
    reference's rule finishes inside the band comes back with EXACTLY its score, every other pair with the optimum (finished
    by the exact tiers on the GPU); the number of pairs finished inside the band is the restatement's.  With CIGARs: every
    alignment valid, cost == reported score, optimum <= score <= the reference rule's (the cost of the returned CIGAR can be
    below the forward score where the band made the search open two gaps back to back: printed they are one gap)."""
    n = 768
    me = 6000
    buf, meta = wfagpu.generate_pairs_model(n, 10000, seed=6, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02,
                                            long_min=30, long_max=150, cluster=0.3)
    batch = aligner.upload(buf, meta)
    exact, _ = aligner.align(batch, pen, max_error=me, compute_cigar=False)
    so, _ = _truth(buf, meta[:32], pen)
    assert np.array_equal(exact[:32], so)
    sr = oracle_lib.band_ref_batch(buf, meta, pen, beta, lam, me, nthreads=16)
    want = np.where(sr >= 0, sr, exact)
    # (pairs within a few scores of the step limit: the reference counts gap-capable steps, this build scores)
    safe = (sr < 0) | (sr < me - 8)
    s0, _ = aligner.align(batch, pen, max_error=me, compute_cigar=False, band=lam, band_width=beta)
    st = aligner.stats()
    assert np.array_equal(s0[safe], want[safe]), np.nonzero(s0 != want)[0][:8]
    assert abs(int(st.pairs_banded) - int((sr >= 0).sum())) <= int((~safe).sum())
    recall = float(((sr >= 0) & (sr == exact)).mean())
    print(f"banded(hard) pen={pen} beta={beta} lambda={lam}: inside the band {st.pairs_banded}/{n}, optimal inside the band {recall:.4f}")
    s, c = aligner.align(batch, pen, max_error=me, compute_cigar=True, band=lam, band_width=beta)
    s2, c2 = aligner.align(batch, pen, max_error=me, compute_cigar=True, band=lam, band_width=beta)
    assert np.array_equal(s, s2) and c == c2
    assert (s >= exact).all() and (s[safe] <= want[safe]).all()
    pairs = wfagpu.pairs_from_layout(buf, meta)
    for i in range(0, n, 3):
        ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], c[i], pen)
        assert ok and cost == s[i]


@pytest.mark.parametrize("min_tier", [0, 1, 2])
def test_adaptive_band_matches_the_reference_rule_on_every_banded_tier(min_tier):
    """One, four and sixteen wavefronts per alignment (tuning.min_tier), short and long pairs, narrow bands (many re-centring
    jumps beyond the rows' guard zones: the range-checked cells), the reference's real HiFi-shaped test pairs: score-only
    results equal the reference rule's restatement pair by pair."""
    hifi = wfagpu.read_seq_file(os.path.join(os.path.dirname(__file__), "golden", "test_hifi.seq"))
    sets = [(wfagpu.layout_pairs(hifi), 3000, (32, 128)),
            (wfagpu.generate_pairs_model(512, 1000, seed=8, error=0.08, indel_frac=0.6, indel_mean=2.5, long_frac=0.03, long_min=10, long_max=60, cluster=0.3), 600, (16, 48, 100)),
            (wfagpu.generate_pairs_model(96, 6000, seed=9, error=0.10, indel_frac=0.7, indel_mean=3.0, long_frac=0.05, long_min=50, long_max=400, cluster=0.5), 9000, (352,))]
    al = wfagpu.DeviceAligner(0, force_band=1, min_tier=min_tier)
    try:
        for (buf, meta), me, betas in sets:
            batch = al.upload(buf, meta)
            for pen in ((2, 3, 1), (1, 2, 1)):
                exact, _ = al.align(batch, pen, max_error=me, compute_cigar=False)
                for beta in betas:
                    for lam in (10, 25):
                        sr = oracle_lib.band_ref_batch(buf, meta, pen, beta, lam, me, nthreads=16)
                        safe = (sr < 0) | (sr < me - 8)
                        want = np.where(sr >= 0, sr, exact)
                        s, _ = al.align(batch, pen, max_error=me, compute_cigar=False, band=lam, band_width=beta)
                        assert np.array_equal(s[safe], want[safe]), (min_tier, pen, beta, lam, np.nonzero(s != want)[0][:8])
            # penalty sets whose scores do not all have a wavefront (e > 1; a common factor): valid, never below the optimum
            for pen in ((5, 3, 2), (4, 6, 2)):
                exact, _ = al.align(batch, pen, max_error=me * 3, compute_cigar=False)
                s, c = al.align(batch, pen, max_error=me * 3, compute_cigar=True, band=25, band_width=betas[-1])
                assert (s >= exact).all()
                pairs = wfagpu.pairs_from_layout(buf, meta)
                for i in range(0, len(pairs), 7):
                    ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], c[i], pen)
                    assert ok and cost == s[i]
            del batch
    finally:
        al.close()


def test_long_non_acgt_pair_backtrace_through_global_scratch(aligner):
    """One wavefront per alignment keeps the op list and the CIGAR text in LDS when they fit next to the staged sequences;
    byte-compare (non-ACGT) pairs of 21 kbp leave no room (2 x 21 KB of sequence bytes), so both go through the global
    scratch and the text is written by a second replay -- the same answers, compared with the checker."""
    rng = random.Random(4)
    pairs = []
    for L in (21000, 20500):
        t = bytearray(rng.choice(b"ACGT") for _ in range(L))
        for pos in rng.sample(range(L), 40):
            t[pos] = ord("N")
        p = bytearray(t)
        for _ in range(L // 50):
            a = rng.randrange(len(p)); r = rng.random()
            if r < 0.4:
                p[a] = rng.choice(b"ACGTN")
            elif r < 0.7:
                del p[a:a + rng.randint(1, 4)]
            else:
                p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 4)))
        pairs.append((bytes(p), bytes(t)))
    pairs.append((b"ACGTNACGT" * 20, b"ACGTNACGA" * 20))
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, (2, 3, 1))
    s, c = _run(aligner, buf, meta, (2, 3, 1), max_error=4000)
    assert aligner.stats().pairs_raw == 3
    assert np.array_equal(s, so) and c == co


@pytest.mark.parametrize("pen,max_error", [((4, 6, 2), 400), ((5, 3, 2), 500), ((3, 1, 4), 600), ((1, 0, 1), 150), ((6, 9, 3), 700)])
def test_penalty_sets_at_scale(aligner, pen, max_error):
    """Other penalty sets in the regime the headline runs in (>= 8192 pairs: sampled per-pair budgets + the lean score loop).
    Gap extensions > 1 take the second form of the lean loop (limits from the row book, scores without a wavefront skipped);
    sets with a common factor -- (4,6,2) = 2 x (2,3,1), (6,9,3) = 3 x (2,3,1) -- run as the reduced set with the scores scaled
    back.  12k pairs of two lengths, scores and CIGARs byte-identical to the oracle's, score-only mode included."""
    buf, meta = _concat_layouts([wfagpu.generate_pairs(8000, 600, 0.05, seed=301), wfagpu.generate_pairs(4000, 900, 0.08, seed=302)])
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
    batch = aligner.upload(buf, meta)
    s, c = aligner.align(batch, pen, max_error=max_error, compute_cigar=True)
    assert aligner.stats().auto_budget > 0
    assert np.array_equal(s, so)
    assert c == co
    s2, _ = aligner.align(batch, pen, max_error=max_error, compute_cigar=False)
    assert np.array_equal(s2, so)


def test_band_is_only_used_where_it_pays():
    """-B on a big batch: the sample that tunes the score budgets runs exactly; when the budgets leave the exact wavefronts
    narrower than 1.75 bands the exact kernels are at least as fast as the band and are used instead (optimal results, no
    pair counted as banded); wfagpu_amd_tuning_t::force_band keeps the band.  Both ways: valid alignments, cost == score >= optimum."""
    buf, meta = wfagpu.generate_pairs(9000, 1500, 0.04, seed=401)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=16)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    for force in (False, True):
        al = wfagpu.DeviceAligner(0, force_band=1 if force else 0)
        try:
            batch = al.upload(buf, meta)
            s, c = al.align(batch, (2, 3, 1), max_error=450, compute_cigar=True, band=25, band_width=128)
            st = al.stats()
            if force:
                assert st.pairs_banded > 8000
                assert (s >= so).all()
                for i in range(0, len(pairs), 9):
                    ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], c[i], (2, 3, 1))
                    assert ok and cost == s[i]
            else:
                assert st.pairs_banded == 0 and st.auto_budget > 0
                assert np.array_equal(s, so) and c == co
        finally:
            al.close()


def test_big_batch_with_budget_misses_and_escalation():
    """A big batch through the chained first pass (wavefront kernel -> compaction -> re-run of the budget misses with its
    length read on the device -> backtrace, one synchronisation): 150k x 300 bp pairs, most at 4 % error, every 97th at
    25 % -- those miss the auto-tuned budgets -- and every 1000th pair unrelated: those exceed max_error too and are
    escalated in a further chain."""
    n = 150_000
    buf, meta = wfagpu.generate_pairs(n, 300, 0.04, seed=77)
    hard, mh = wfagpu.generate_pairs(n // 97 + 1, 300, 0.25, seed=78)
    wild, mw = wfagpu.generate_pairs(n // 1000 + 1, 300, 0.75, seed=79)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    ph, pw = wfagpu.pairs_from_layout(hard, mh), wfagpu.pairs_from_layout(wild, mw)
    for j, i in enumerate(range(0, n, 97)):
        pairs[i] = ph[j]
    for j, i in enumerate(range(5, n, 1000)):
        pairs[i] = pw[j]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=16)
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        s, c = al.align(batch, (2, 3, 1), max_error=200, compute_cigar=True)
        st = al.stats()
        assert st.auto_budget > 0 and st.pairs_budget_missed > 1000     # the 25 % pairs
        assert st.pairs_retried > st.pairs_budget_missed - 1 and sum(st.pairs_tier) == n
        assert int((so > 200).sum()) > 50                                # pairs beyond max_error: escalated
        assert np.array_equal(s, so)
        assert c == co
        s2, _ = al.align(batch, (2, 3, 1), max_error=200, compute_cigar=False)
        assert np.array_equal(s2, so)
    finally:
        al.close()


def test_cigar_runs_of_six_and_more_digits():
    """Near-identical reads beyond 100 kbp: a match run prints as six or more digits ("150000M").  The text bound is sized
    from the longest sequence of the batch (not from a fixed five digits per item) and the RLE printer handles ten digits."""
    rng = random.Random(99)
    a = bytes(rng.choice(b"ACGT") for _ in range(150_000))
    b = bytearray(a); b[70_000] = ord("A") if a[70_000] != ord("A") else ord("C")
    c = a[:100_000] + a[100_003:]
    pairs = [(a, a), (a, bytes(b)), (a, c), (c, a)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, (2, 3, 1))
    assert co[0] == "150000M" and co[1] == "70000M1X79999M"
    al = wfagpu.DeviceAligner(0)
    try:
        s, cg = _run(al, buf, meta, (2, 3, 1), max_error=100)
        assert np.array_equal(s, so) and cg == co
    finally:
        al.close()


@pytest.mark.parametrize("trace_mode", [1, 2])
def test_fallback_backtrace_paths(trace_mode):
    """wfagpu_amd_tuning_t::trace_mode 1: lane-per-alignment walk + the windowed emit kernel whatever the length (the
    fallback for sequences that leave the wave-per-alignment kernel no LDS); 2: never several alignments per wavefront.
    Same CIGARs as the default choice."""
    rng = random.Random(5150 + trace_mode)
    pairs = _rand_pairs(rng, 200, 2500, err=0.06) + _rand_pairs(rng, 64, 300, err=0.1) + [(b"", b"ACGT"), (b"ACGT" * 700, b"ACGT" * 650)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    al = wfagpu.DeviceAligner(0, trace_mode=trace_mode)
    try:
        s, cg = _run(al, buf, meta, (2, 3, 1), max_error=2000)
        assert np.array_equal(s, so) and cg == co
    finally:
        al.close()


def test_long_alignments_walked_by_wavefronts_and_replayed_by_lanes():
    """Big passes of long alignments (8192 and more; tuning.trace_mode 4: any size): the wave-per-alignment kernel only walks and
    leaves its op lists in slots of the global scratch, the lane kernel replays eight and more alignments per wavefront with
    sequences AND op lists staged in LDS, the texts are compacted.  Ragged lengths, scores from 0 to thousands, an empty
    pattern, a budget most pairs miss (they finish in the wider tiers, in passes of their own); penalties with a costly
    extension.  Same scores and CIGARs as WFA2."""
    rng = random.Random(424242)
    pairs = (_rand_pairs(rng, 150, 4000, err=0.08) + _rand_pairs(rng, 60, 2000, err=0.15) + _rand_pairs(rng, 40, 3000, err=0.0)
             + [(b"", b"ACGT" * 600), (b"ACGT" * 900, b"ACGT" * 650), (b"A" * 3000, b"C" * 3000)])
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen, max_error in (((2, 3, 1), 4000), ((4, 6, 2), 20000), ((2, 3, 1), 700)):
        so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
        al = wfagpu.DeviceAligner(0, trace_mode=4)
        try:
            s, cg = _run(al, buf, meta, pen, max_error=max_error)
            assert np.array_equal(s, so), (pen, max_error)
            assert cg == co, (pen, max_error)
            assert al.stats().pairs_trace_split >= len(pairs) // 2
        finally:
            al.close()


def test_rings_that_fill_lds_to_the_last_bytes():
    """Score budgets swept in small steps across the point where the sixteen-wave ring stops fitting a CU's LDS: every plan on
    either side must launch (a ring within the last 256 bytes of LDS once did not -- a static LDS word had crept into the
    kernels -- and the failed occupancy query left an error behind that the next, perfectly good launch reported: found by
    scratch/soak.py), scores exact."""
    rng = random.Random(9090)
    pairs = _rand_pairs(rng, 6, 2500, err=0.03)
    pairs = [(p, t) for p, t in pairs if len(p) > 2000 and len(t) > 2000][:4] or pairs[:4]
    buf, meta = wfagpu.layout_pairs(pairs)
    pen = (3, 2, 5)
    so, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False, nthreads=4)
    al = wfagpu.DeviceAligner(0, min_tier=2)
    try:
        batch = al.upload(buf, meta)
        tiers = set()
        for max_error in range(18600, 21400, 20):      # (20 scores = 4 diagonals = 160 bytes of ring: no 256-byte window is skipped)
            s, _ = al.align(batch, pen, max_error=max_error, compute_cigar=False)
            assert np.array_equal(s, so), max_error
            tiers.add(int(al.stats().main_launch_tier))
        assert len(tiers) >= 2, tiers      # (the sweep crossed from the whole ring in LDS to the hybrid ring)
    finally:
        al.close()


@pytest.mark.parametrize("min_tier", [0, 1, 2])
def test_packing_while_staging(min_tier):
    """Resident ASCII batches of reads of 512 bases and more have no pack kernel: the wavefront kernels pack the pairs they stage
    (one wave, four waves, sixteen waves per alignment: tuning.min_tier) and send pairs with bytes outside ACGT -- N, lower case,
    in the first word, the last one, in either sequence -- to the byte-compare class.  Same scores, CIGARs and byte-compare
    counts as with the pack kernel (tuning.no_fused_pack), and the checker's."""
    rng = random.Random(8800 + min_tier)
    pairs = _rand_pairs(rng, 700, 1400, err=0.04)
    pairs = [(p, t) for p, t in pairs if max(len(p), len(t)) >= 520][:500]
    def spoil(seq, where):
        b = bytearray(seq)
        if b:
            b[where % len(b)] = rng.choice(b"Nnacgt")
        return bytes(b)
    for i in range(0, len(pairs), 9):
        p, t = pairs[i]
        k = (i // 9) % 4
        pairs[i] = (spoil(p, 0), t) if k == 0 else (p, spoil(t, len(t) - 1)) if k == 1 else (spoil(p, len(p) - 1), spoil(t, 3)) if k == 2 else (p, spoil(t, len(t) // 2))
    pairs += [(b"", b"ACGT" * 150), (b"ACGT" * 150, b"")]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    out, raws, packs = [], [], []
    for no_fused in (0, 1):
        al = wfagpu.DeviceAligner(0, min_tier=min_tier, no_fused_pack=no_fused)
        try:
            out.append(_run(al, buf, meta, (2, 3, 1), max_error=400))
            st = al.stats()
            raws.append(int(st.pairs_raw)); packs.append(float(st.pack_ms))
        finally:
            al.close()
    for s, cg in out:
        assert np.array_equal(s, so) and cg == co
    assert raws[0] == raws[1] and raws[0] >= len(pairs) // 12, raws
    assert packs[0] == 0.0 and packs[1] > 0.0, packs


@pytest.mark.parametrize("cigar", [False, True])
def test_misses_after_a_batch_without_any(cigar):
    """A stream of batches under inherited budgets: the speculative re-run launch of budget misses is left out once a batch had
    no miss at all (short reads under budgets with room); pairs that miss in a LATER batch are then re-run by a chain of their
    own.  Batch 1 tunes the budgets on its sample, batch 2 (the same reads) has no miss, batch 3 holds 1 % of reads at 15 %
    error far beyond their budgets, batch 4 sees those again with the re-run launch back in place: every score (and CIGAR)
    equals the checker's, and the misses are counted."""
    n = 12000
    clean, mc = wfagpu.generate_pairs(n, 150, 0.02, seed=611)
    hard, mh = wfagpu.generate_pairs(n // 100, 150, 0.15, seed=612)
    pairs = wfagpu.pairs_from_layout(clean, mc)
    mixed = list(pairs)
    for j, p in enumerate(wfagpu.pairs_from_layout(hard, mh)):
        mixed[97 * j + 5] = p
    al = wfagpu.DeviceAligner(0)
    try:
        missed = []
        for batch_pairs in (pairs, pairs, mixed, mixed):
            buf, meta = wfagpu.layout_pairs(batch_pairs)
            so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=cigar, nthreads=8)
            s, cg = _run(al, buf, meta, (2, 3, 1), max_error=60, cigar=cigar)
            st = al.stats()
            assert np.array_equal(s, so)
            if cigar:
                assert cg == co
            assert st.auto_budget > 0
            missed.append(int(st.pairs_budget_missed))
            al.hint_same_stream(True)
        assert missed[1] == 0 and missed[2] >= n // 200 and missed[3] == missed[2], missed
    finally:
        al.close()


@pytest.mark.parametrize("max_error,pen", [(40, (2, 3, 1)), (124, (2, 3, 1)), (125, (2, 3, 1)), (90, (4, 6, 2)), (60, (3, 1, 4))])
def test_one_kernel_backtrace_of_short_alignments(max_error, pen):
    """Chains whose scores are bounded by 124 walk and replay their alignments in ONE kernel (wfa_trace_lane_kernel: op lists in
    LDS, one text allocation per wavefront); above that bound, and with tuning.trace_mode = 3, walk + emit + compaction run as
    before.  Same scores and CIGARs either way, and the checker's: short reads over ACGT and with other bytes (byte-compare
    class), pairs beyond the budget (re-run with a wider one in another chain), empty sequences, a batch big enough for the
    scratch + compaction path of the three-kernel form."""
    rng = random.Random(4100 + max_error)
    pairs = _rand_pairs(rng, 9000, 160, err=0.03) + _rand_pairs(rng, 300, 200, err=0.12)
    pairs += _rand_pairs(rng, 200, 120, alphabet=b"ACGTN", err=0.04) + [(b"", b"ACGT"), (b"", b""), (b"ACGTNNNN", b""), (b"A" * 150, b"C" * 150)]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
    out = []
    for mode in (0, 3):
        al = wfagpu.DeviceAligner(0, trace_mode=mode)
        try:
            out.append(_run(al, buf, meta, pen, max_error=max_error))
        finally:
            al.close()
    assert np.array_equal(out[0][0], so) and out[0][1] == co
    assert np.array_equal(out[1][0], so) and out[1][1] == co


@pytest.mark.parametrize("pen,expect_short", [((2, 3, 1), True), ((4, 6, 2), True), ((1, 2, 1), True), ((1, 0, 1), True), ((4, 6, 1), True),
                                              ((3, 4, 1), True), ((7, 7, 1), True), ((8, 2, 1), True), ((6, 9, 3), True),
                                              ((5, 3, 2), True), ((3, 1, 4), True), ((7, 2, 3), True), ((3, 5, 2), True), ((1, 0, 2), True), ((8, 4, 4), True),
                                              ((2, 0, 3), True), ((3, 1, 5), False), ((9, 2, 1), False), ((2, 8, 1), False)])
def test_short_wavefront_tier_scores(pen, expect_short):
    """Score-only batches of short reads run on tier 5 (short_kernel.hip: four or two alignments per wavefront, rings in
    registers, neighbours by DPP row shifts) once their budgets are tuned: 20k x 150 bp pairs at 2 %, with every 200th pair at
    12 % (below the 99th percentile the budgets come from: it misses its budget and is re-run in the ordinary tiers), pairs whose lengths differ so much that their diagonal window does
    not fit a group (BAND failure -> ordinary tiers), empty and one-base sequences.  Every score equals the checker's.  The tier
    is compiled for every penalty set with e <= 4 and max(x, o + e) <= 8 once a common factor is divided out (rings of up to
    eight M -- and, for e > 1, I and D -- registers per lane; the reference's own test sets (5,3,2), (3,5,2), (3,1,4):
    tests/test_api.c:59-219); other sets take the ordinary path."""
    n = 20000
    buf, meta = wfagpu.generate_pairs(n, 150, 0.02, seed=515)
    hard, mh = wfagpu.generate_pairs(n // 200 + 1, 150, 0.12, seed=516)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    ph = wfagpu.pairs_from_layout(hard, mh)
    for j, i in enumerate(range(3, n, 200)):      # (odd positions: the strided sample of the budget tuning never draws one)
        pairs[i] = ph[j]
    rng = random.Random(517)
    for i in range(7, n, 400):          # lengths far apart: |kend| beyond any 32-lane window
        t = bytes(rng.choice(b"ACGT") for _ in range(150))
        pairs[i] = (t[:rng.randint(20, 90)], t)
    pairs[11] = (b"", b"ACGT"); pairs[12] = (b"A", b"A"); pairs[13] = (b"", b""); pairs[14] = (b"ACGTACGT", b"")
    buf, meta = wfagpu.layout_pairs(pairs)
    so, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False, nthreads=16)
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        for max_error in (45 * max(1, pen[0] // 2), 400):
            s, _ = al.align(batch, pen, max_error=max_error, compute_cigar=False)
            st = al.stats()
            assert np.array_equal(s, so), (pen, max_error)
            if expect_short:
                assert st.pairs_tier[5] > n * 0.9 and st.pairs_retried > 100, list(st.pairs_tier)
            else:
                assert st.pairs_tier[5] == 0
        # and with CIGARs: the same tier (one row of origin bytes per score and group), every CIGAR byte-identical to WFA2's
        _, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=16)
        for max_error in (45 * max(1, pen[0] // 2), 400):
            s2, c2 = al.align(batch, pen, max_error=max_error, compute_cigar=True)
            st = al.stats()
            assert np.array_equal(s2, so) and c2 == co, (pen, max_error)
            if expect_short:
                assert st.pairs_tier[5] > n * 0.9, list(st.pairs_tier)
            else:
                assert st.pairs_tier[5] == 0
        # the A/B switch: CIGAR calls on the ordinary tiers, same answers
        al.set_tuning(no_short_cigar=1)
        s3, c3 = al.align(batch, pen, max_error=400, compute_cigar=True)
        assert np.array_equal(s3, so) and c3 == co and al.stats().pairs_tier[5] == 0
        al.set_tuning()
    finally:
        al.close()
