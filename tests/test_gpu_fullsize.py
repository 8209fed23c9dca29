"""BASELINE.json configs[1], [2], [3] and [4] at FULL size through the C-ABI (wfagpu_amd_align_device), compared with the
reference's WFA2 (oracle/_ref, or the C restatement when that is not there):

  configs[1]  100k x 150 bp @ 2 %, SCORE-ONLY, max_error 45 -- exactly what `bench.py --workload cfg2` times: the
              several-alignments-per-wavefront tier (short_kernel.hip), every one of the 100 000 scores;

  configs[2]  1M x 1 kbp @ 5 %: with the arena the library picks ALL 1M scores AND CIGAR strings; with an arena cap that
              forces the batch through several arena-bound passes a 100k-pair stratified sample that contains EVERY pair
              that missed its auto-tuned budget (the re-run path; the sample covers every pass);
  configs[3]  16 384 x 10 kbp @ 3 % (the shape that picks the four-wave tier with six rings per CU): every score, EVERY CIGAR
              string (round 6; 256 through round 5), every CIGAR valid with cost == score -- exact; with -B auto -t 512 the band policy takes the
              banded kernels: valid, cost == score, optimum <= score <= the reference band rule's;
  configs[4]  1 024 x 30 kbp @ 10 % (hybrid ring tier, one workgroup per CU): every score, CIGAR identity on 128 pairs,
              every CIGAR valid with cost == score.
"""
import os

import numpy as np
import pytest

import oracle_lib
import wfagpu

pytestmark = pytest.mark.gpu
PEN = (2, 3, 1)


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 32)


def _truth(buf, meta, cigar):
    if oracle_lib.have_ref():
        return oracle_lib.ref_batch(buf, meta, PEN, cigar=cigar, memory_mode=0, nthreads=_threads())
    s, c, _ = oracle_lib.oracle_batch(buf, meta, PEN, cigar=cigar, nthreads=_threads())
    return s, c


def _device_results(al, batch, max_error, band=-1, band_width=0):
    d_scores, ptrs = al.align(batch, PEN, max_error=max_error, compute_cigar=True, band=band, band_width=band_width, fetch=False)
    st = al.stats()
    n = batch.num_pairs
    scores = d_scores.cpu().numpy()
    off = wfagpu._d2h(ptrs[1], 8 * n).view(np.uint64)
    ln = wfagpu._d2h(ptrs[2], 4 * n).view(np.uint32)
    text = wfagpu._d2h(ptrs[0], int(st.text_bytes)).tobytes()
    assert not (ln == 0xFFFFFFFF).any()
    return scores, off, ln, text, st


def _cigar(text, off, ln, i):
    o = int(off[i])
    return text[o:o + int(ln[i])].decode()


def _launch(buf, meta, max_error, **launch_cfg):
    """launch_alignments() on host buffers (one batch = the whole call, as the CLI's default): scores, CIGAR strings, stats."""
    import ctypes as C
    lib = wfagpu.load()
    n = len(meta)
    res = C.POINTER(wfagpu.AlignmentResult)()
    assert lib.initialize_wfa_results(C.byref(res), n, 256)
    opt = wfagpu.Options(max_error=max_error, threads_per_block=64, num_workers=0, band=-1, batch_size=n, num_alignments=n,
                         penalties=wfagpu.Penalties(*PEN), compute_cigar=True)
    wfagpu.configure_launch(**launch_cfg)
    m2 = meta.copy()
    try:
        lib.launch_alignments(buf.ctypes.data, buf.nbytes, m2.ctypes.data, res, opt, False)
        st = wfagpu.last_launch_stats()
        scores = np.array([res[i].error for i in range(n)], dtype=np.int64)
        cigars = [C.string_at(res[i].cigar.buffer).decode() for i in range(n)]
    finally:
        lib.destroy_wfa_results(res, n)
        wfagpu.configure_launch()
        lib.wfagpu_amd_release_cache()
    return scores, cigars, st


def test_cfg2_full_size_score_only_as_benched():
    """configs[1] in its own mode at its own size: score-only, max_error 45 (the CLI's automatic value for 150 bp reads),
    budgets tuned on the strided sample, the whole batch on tier 5, then a second call of the same stream that inherits
    the budgets (what the timed steps of bench.py do).  ALL scores against the checker."""
    n = 100_000
    buf, meta = wfagpu.generate_pairs(n, 150, 0.02, seed=1000, nthreads=_threads())
    so, _ = _truth(buf, meta, cigar=False)
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        for call in range(2):
            d_scores, _ = al.align(batch, PEN, max_error=45, compute_cigar=False, fetch=False)
            st = al.stats()
            scores = d_scores.cpu().numpy()
            assert np.array_equal(scores, so), call
            assert st.auto_budget > 0
            # first call: the 4096-pair sample the budgets are tuned on keeps its alignments (one-wave tier), everybody else
            # runs on tier 5; second call (budgets inherited, the timed steps of bench.py): everybody -- but for budget
            # misses, which the one-wave tier re-runs
            assert st.pairs_tier[5] + st.pairs_tier[0] == n, list(st.pairs_tier)
            assert st.pairs_tier[0] <= (4096 if call == 0 else 0) + st.pairs_budget_missed, list(st.pairs_tier)
            assert st.pairs_tier[5] >= (n - 4096 if call == 0 else 0.99 * n), list(st.pairs_tier)
            # (the longest wavefront launch of the first call may be the one-wave launch of the 4096 sampled pairs: 0.13 ms,
            # against 0.11 ms for the other 95 904 pairs on tier 5)
            assert st.main_launch_tier == 5 or call == 0
            al.hint_same_stream(True)
    finally:
        al.close()


@pytest.mark.parametrize("arena_limit_gib", [0, 2])
def test_cfg3_full_size_every_budget_miss_and_every_pass(arena_limit_gib):
    n = 1_000_000
    buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=_threads())
    al = wfagpu.DeviceAligner(0, arena_limit_bytes=arena_limit_gib << 30)
    try:
        batch = al.upload(buf, meta)
        scores, off, ln, text, st = _device_results(al, batch, max_error=300)
    finally:
        al.close()
    assert st.auto_budget > 0 and st.pairs_budget_missed > 0
    if arena_limit_gib:
        assert st.sub_batches > 1          # (passes take contiguous index ranges: the strided sample covers each of them)
    # every pair that missed its budget: budget_i = q * len_i / 1024 + 2 with len_i the longer sequence of the pair and
    # st.auto_budget the value for the longest pair of the batch -- a per-pair LOWER bound of it selects a superset
    lens = np.maximum(meta["pattern_len"], meta["text_len"]).astype(np.int64)
    budget_lo = (st.auto_budget - 2) * lens // int(lens.max())
    missed = np.nonzero(scores > budget_lo)[0]
    assert st.pairs_budget_missed <= len(missed) <= 20 * st.pairs_budget_missed
    # arena picked by the library: ALL 1M scores and CIGAR strings against the reference's WFA2 (16 s of its time on the box's
    # 16 cores), in slices of 100k pairs; several arena-bound passes: the stratified sample + every budget miss
    sample = np.arange(n) if arena_limit_gib == 0 else np.union1d(np.arange(0, n, 10), missed)
    assert len(sample) == n or 100_000 <= len(sample) <= 150_000
    for lo in range(0, len(sample), 100_000):
        part = sample[lo:lo + 100_000]
        so, co = _truth(buf, meta[part], cigar=True)
        assert np.array_equal(scores[part], so), lo
        bad = [int(i) for j, i in enumerate(part) if _cigar(text, off, ln, i) != co[j]]
        assert not bad, bad[:5]
    # size-independent property on everything: the text arena is dense and every text ends where the next begins
    order = np.argsort(off)
    assert np.array_equal(off[order][1:], (off[order] + ln[order] + 1)[:-1])
    assert int(off[order][-1] + ln[order][-1] + 1) == st.text_bytes


@pytest.mark.parametrize("band", [None, (25, 512)])
def test_cfg4_full_size_scores_and_cigars(band):
    n = 16_384
    buf, meta = wfagpu.generate_pairs(n, 10_000, 0.03, seed=1000, nthreads=_threads())
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        scores, off, ln, text, st = _device_results(al, batch, max_error=3000, band=band[0] if band else -1,
                                                    band_width=band[1] if band else 0)
    finally:
        al.close()
    so, _ = _truth(buf, meta, cigar=False)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    if band is not None:
        # The band is a permission to approximate: the sampled budgets leave the exact wavefronts of this
        # batch 1.94 bands wide, and since round 5 the banded kernels are the faster way through such pairs: the policy takes the
        # band (the sample and what passes max_error stay exact).  Every alignment valid, cost == score, optimum <= score <= the
        # reference band rule's score.
        assert st.pairs_banded > 0.9 * n and st.auto_budget > 0
        sr = oracle_lib.band_ref_batch(buf, meta, PEN, band[1], band[0], 3000, nthreads=_threads())
        want = np.where(sr >= 0, sr, so)
        assert (scores >= so).all() and (scores <= want).all()
        print(f"cfg4 by policy: {st.pairs_banded}/{n} pairs banded, recall {(scores == so).mean():.4f}")
        for i in range(n):
            ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], _cigar(text, off, ln, i), PEN)
            assert ok and cost == scores[i], i
        return
    assert st.pairs_banded == 0
    assert st.pairs_tier[1] > n // 2, list(st.pairs_tier)      # the four-wave tier
    assert np.array_equal(scores, so)
    # CIGAR strings of ALL 16 384 pairs against the reference's WFA2 (round 6; 256 of them through round 5 -- the reference's -c path
    # checks every pair: lib/align.cu:258-326; ~5 s on the box's 16 cores)
    _, co = _truth(buf, meta, cigar=True)
    got = [_cigar(text, off, ln, i) for i in range(n)]
    assert got == co, next(i for i in range(n) if got[i] != co[i])
    for i in range(n):
        ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], got[i], PEN)
        assert ok and cost == scores[i], i
    if band is None:
        # the same pairs host to host: a call of few long reads (330 MB) is pipelined in batches of >= 8192 pairs, with the
        # sequences packed on the host or going up as ASCII -- every score and every CIGAR string as from the resident batch
        for host_pack in (1, -1):
            s2, c2, lst = _launch(buf, meta, 3000, host_pack=host_pack)
            assert lst["batches"] == 2 and lst["host_packed_batches"] == (2 if host_pack > 0 else 0)
            assert np.array_equal(s2, scores)
            assert c2 == [_cigar(text, off, ln, i) for i in range(n)]


def test_cfg4_full_size_with_the_band_forced():
    """configs[3] as BASELINE states it -- 10 kbp pairs at 3 %, -B auto (re-centre every 25 scores) -t 512 -- on the BANDED
    kernels at full size (tuning.force_band: by default this batch runs exactly, the band policy of DESIGN.md section 8):
    score-only results equal the reference's adaptive-band rule restated on the CPU (oracle/band_oracle.c) for ALL 16 384
    pairs; with CIGARs every alignment is valid, its cost is the reported score, optimum <= score <= the reference rule's."""
    n = 16_384
    buf, meta = wfagpu.generate_pairs(n, 10_000, 0.03, seed=1000, nthreads=_threads())
    so, _ = _truth(buf, meta, cigar=False)
    sr = oracle_lib.band_ref_batch(buf, meta, PEN, 512, 25, 3000, nthreads=_threads())
    assert (sr >= 0).all()                      # (the band of 512 diagonals holds every one of these pairs)
    al = wfagpu.DeviceAligner(0, force_band=1)
    try:
        batch = al.upload(buf, meta)
        d_s, _ = al.align(batch, PEN, max_error=3000, compute_cigar=False, band=25, band_width=512, fetch=False)
        st0 = al.stats()
        s0 = d_s.cpu().numpy().copy()
        scores, off, ln, text, st = _device_results(al, batch, max_error=3000, band=25, band_width=512)
    finally:
        al.close()
    # (the rest: the 1024-pair sample the score budgets are tuned on -- aligned exactly -- and budget misses, re-run exactly)
    assert st0.pairs_banded > 0.9 * n and st.pairs_banded > 0.9 * n
    assert np.array_equal(s0, sr)
    print(f"cfg4, band forced: {st.pairs_banded}/{n} pairs inside the band, recall {(sr == so).mean():.4f} (reference rule: the same pairs)")
    assert (scores >= so).all() and (scores <= sr).all()
    # With CIGARs the reported score is the cost of the CIGAR that is returned.  It differs from the forward score of the band
    # rule only where the banded search closed a gap and opened the same kind again right behind it (printed, the two are ONE
    # gap): each such place saves exactly one gap opening.  Counted and bounded, every other pair must carry the rule's score.
    fixed = np.nonzero(scores != sr)[0]
    saved = (sr - scores)[fixed]
    print(f"cfg4, band forced, with CIGARs: {len(fixed)} of {n} pairs report a cost below the rule's forward score "
          f"(by {int(saved.min()) if len(fixed) else 0}..{int(saved.max()) if len(fixed) else 0}); all others equal it")
    assert (saved > 0).all() and (saved % PEN[1] == 0).all() and len(fixed) <= n // 20
    pairs = wfagpu.pairs_from_layout(buf, meta)
    for i in range(n):
        ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], _cigar(text, off, ln, i), PEN)
        assert ok and cost == scores[i], i


def test_cfg5_full_size_scores_and_cigars():
    n = 1024
    buf, meta = wfagpu.generate_pairs(n, 30_000, 0.10, seed=1000, nthreads=_threads())
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        scores, off, ln, text, st = _device_results(al, batch, max_error=9000)
    finally:
        al.close()
    assert st.pairs_tier[4] > n // 2, list(st.pairs_tier)      # the hybrid ring tier
    so, _ = _truth(buf, meta, cigar=False)
    assert np.array_equal(scores, so)
    # CIGAR strings of 128 of the 1024 pairs against the reference's WFA2 (32 through round 5; ~0.15 s per pair and core)
    idx = np.arange(0, n, 8)[:128]
    _, co = _truth(buf, meta[idx], cigar=True)
    assert [_cigar(text, off, ln, i) for i in idx] == co
    pairs = wfagpu.pairs_from_layout(buf, meta)
    for i in range(n):
        ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], _cigar(text, off, ln, i), PEN)
        assert ok and cost == scores[i], i
