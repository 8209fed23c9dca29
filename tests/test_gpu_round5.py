"""Round-5 additions, through the C-ABI on the GPU:
  * the one-wave wavefront kernels walking their own alignments back (align/walk_epilogue.inc, tuning.kernel_walk: an option, off by
    default -- it loses 0.9 ms per 1M-pair step, EXPERIMENTS.md): same scores and CIGAR strings as with wfa_walk_kernel and as the
    checker, incl. byte-compare pairs, budget misses, arena-bound passes and mixed chains;
  * every wavefronts-per-alignment choice of the banded kernels (one, two, four, sixteen) gives the reference band rule's scores;
  * bench.py as a torch.distributed job of ONE rank (self-spawned torchrun child, RCCL process group, barrier + all-reduce +
    all-gather executed): the N > 1 harness on the one GPU there is;
  * one launch_alignments() call of 1M pairs sharded over eight device slots returns what the one-slot call returns.
"""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib
import wfagpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEN = (2, 3, 1)


def _truth(buf, meta, pen, cigar=True):
    if oracle_lib.have_ref():
        return oracle_lib.ref_batch(buf, meta, pen, cigar=cigar, memory_mode=0, nthreads=16)
    s, c, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=cigar, nthreads=16)
    return s, c


@pytest.mark.parametrize("pen", [(2, 3, 1), (4, 6, 2), (3, 1, 4)])
def test_wavefront_kernels_walk_their_own_alignments(pen):
    """12 000 pairs of 1.2 kbp at 6 % (scores well above 124: the lane-per-alignment replay), a handful with bytes outside ACGT
    (the byte-compare class: longer staged sequences, the wave-per-alignment backtrace) and a sub-population at 20 % that misses the
    tuned budgets (re-run link of the chain: walked as well).  tuning.kernel_walk = 0 is the default path (wfa_walk_kernel): both must give the checker's scores and CIGAR strings."""
    rng = random.Random(5 + sum(pen))
    buf_a, meta_a = wfagpu.generate_pairs(11000, 1200, 0.06, seed=31)
    pairs = wfagpu.pairs_from_layout(buf_a, meta_a)
    buf_b, meta_b = wfagpu.generate_pairs(900, 700, 0.20, seed=32)
    pairs += wfagpu.pairs_from_layout(buf_b, meta_b)
    for _ in range(100):
        p, t = pairs[rng.randrange(11000)]
        p = bytearray(p)
        for pos in rng.sample(range(len(p)), 3):
            p[pos] = rng.choice(b"NnRY")
        pairs.append((bytes(p), t))
    rng.shuffle(pairs)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, pen)
    me = int(so.max()) + 10
    got = {}
    for kw in (1, 0):
        al = wfagpu.DeviceAligner(0, kernel_walk=kw)
        try:
            batch = al.upload(buf, meta)
            s, c = al.align(batch, pen, max_error=me, compute_cigar=True)
            st = al.stats()
        finally:
            al.close()
        assert np.array_equal(s, so), kw
        assert c == co, kw
        got[kw] = st
    # (round 6: the option went with the row table it read -- the one-wave tier keeps its origin bytes in tiles, which make
    # wfa_walk_kernel itself cheap; the switch is accepted and ignored)
    assert got[1].pairs_walked_in_kernel == 0 and got[0].pairs_walked_in_kernel == 0
    assert got[1].pairs_budget_missed > 0 and got[1].pairs_raw == 100


def test_kernel_walk_with_a_small_arena_and_mixed_tiers():
    """The op list of a walked alignment comes out of the workgroup's arena chunk: an arena that holds a fraction of the batch (several
    passes, NOMEM pairs re-queued -- some of them at the moment the op list is claimed) and a chain whose second link runs on a
    multi-wave tier (not walked in the kernel: wfa_walk_kernel runs for those and skips the pairs that were)."""
    buf_a, meta_a = wfagpu.generate_pairs(9000, 1000, 0.05, seed=41)
    pairs = wfagpu.pairs_from_layout(buf_a, meta_a)
    buf_b, meta_b = wfagpu.generate_pairs(300, 1000, 0.30, seed=42)      # far beyond the tuned budgets: re-run with a wide window
    pairs += wfagpu.pairs_from_layout(buf_b, meta_b)
    random.Random(9).shuffle(pairs)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, PEN)
    al = wfagpu.DeviceAligner(0, arena_bytes=48 << 20, kernel_walk=1)
    try:
        batch = al.upload(buf, meta)
        s, c = al.align(batch, PEN, max_error=int(so.max()) + 5, compute_cigar=True)
        st = al.stats()
    finally:
        al.close()
    assert st.sub_batches > 1
    assert np.array_equal(s, so) and c == co


@pytest.mark.parametrize("band_tier", [1, 2, 3, 4])
def test_banded_kernels_at_every_wavefront_count(band_tier):
    """One, two, four and sixteen wavefronts per alignment (tuning.band_tier): score-only results equal the reference band rule's
    restatement pair by pair, on pairs whose band re-centres often (lambda 10) and rarely (lambda 100); with CIGARs: valid, cost ==
    score, never below the optimum, never above the rule's."""
    buf, meta = wfagpu.generate_pairs_model(160, 5000, seed=21, error=0.08, indel_frac=0.6, indel_mean=2.5, long_frac=0.03, long_min=20, long_max=120, cluster=0.3)
    me = 4000
    al = wfagpu.DeviceAligner(0, force_band=1, band_tier=band_tier)
    try:
        batch = al.upload(buf, meta)
        exact, _ = al.align(batch, PEN, max_error=me, compute_cigar=False)
        for beta in (96, 352):
            for lam in (10, 100):
                sr = oracle_lib.band_ref_batch(buf, meta, PEN, beta, lam, me, nthreads=16)
                safe = (sr < 0) | (sr < me - 8)
                want = np.where(sr >= 0, sr, exact)
                s, _ = al.align(batch, PEN, max_error=me, compute_cigar=False, band=lam, band_width=beta)
                assert np.array_equal(s[safe], want[safe]), (band_tier, beta, lam, np.nonzero(s != want)[0][:8])
                assert al.stats().pairs_banded == int((sr >= 0).sum()) or not safe.all()
        s, c = al.align(batch, PEN, max_error=me, compute_cigar=True, band=25, band_width=352)
        sr = oracle_lib.band_ref_batch(buf, meta, PEN, 352, 25, me, nthreads=16)
        want = np.where(sr >= 0, sr, exact)
        assert (s >= exact).all() and (s <= want).all()
        pairs = wfagpu.pairs_from_layout(buf, meta)
        for i in range(len(pairs)):
            ok, cost = oracle_lib.check_cigar(pairs[i][0], pairs[i][1], c[i], PEN)
            assert ok and cost == s[i], i
    finally:
        al.close()


def test_bench_harness_runs_as_a_one_rank_rccl_job():
    """`bench.py --gpus 1 --force-dist`: the script starts its own `python -m torch.distributed.run` child with one rank, the rank
    initialises the RCCL process group (backend "nccl") on its device and times the steps between two barriers, the max-over-ranks
    all-reduce and the all-gather of the per-rank clocks run on the GPU -- everything the N > 1 runs do except a second device."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--workload", "cfg2", "--steps", "50", "--warmup", "3",
                        "--no-cpu-baseline", "--no-host-to-host", "--no-configs"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 1 and out["steps"] == 50 and out["value"] > 1e6
    assert "rccl" in (out["config"]["process_group"] or "")
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["ms_per_step"] <= out["ms_per_step"] * 1.01
    assert out["parity_sample"]["bit_exact_vs_oracle"] is True


def test_one_million_pairs_sharded_over_eight_device_slots():
    """launch_alignments() on 1M x 1 kbp pairs (BASELINE configs[2]) cut by the library over EIGHT device slots (virtual devices: own
    threads, lanes, streams and arenas each, all on the one GPU): every score and every CIGAR string equals the one-slot call's, and a
    sample of them the checker's."""
    import ctypes as C
    n = 1_000_000
    buf, meta = wfagpu.generate_pairs(n, 1000, 0.05, seed=1000, nthreads=16)
    lib = wfagpu.load()

    def call(**cfg):
        res = C.POINTER(wfagpu.AlignmentResult)()
        assert lib.initialize_wfa_results(C.byref(res), n, 256)
        opt = wfagpu.Options(max_error=300, threads_per_block=64, num_workers=0, band=-1, batch_size=n, num_alignments=n,
                             penalties=wfagpu.Penalties(*PEN), compute_cigar=True)
        wfagpu.configure_launch(**cfg)
        m2 = meta.copy()
        try:
            lib.launch_alignments(buf.ctypes.data, buf.nbytes, m2.ctypes.data, res, opt, False)
            st = wfagpu.last_launch_stats()
            scores = np.fromiter((res[i].error for i in range(n)), dtype=np.int64, count=n)
            cigars = [C.string_at(res[i].cigar.buffer) for i in range(n)]
        finally:
            lib.destroy_wfa_results(res, n)
            wfagpu.configure_launch()
            lib.wfagpu_amd_release_cache()
        return scores, cigars, st

    s1, c1, st1 = call(num_devices=1)
    s8, c8, st8 = call(virtual_devices=8)
    assert st1["devices"] == 1 and st8["devices"] == 8
    assert np.array_equal(s1, s8)
    assert c1 == c8
    idx = np.arange(0, n, 500)
    so, co = _truth(buf, meta[idx], PEN)
    assert np.array_equal(s8[idx], so)
    assert [c8[i].decode() for i in idx] == co


@pytest.mark.parametrize("cigar", [False, True])
def test_short_reads_packed_by_their_own_tier(cigar):
    """Reads of 150 bases are packed by tier 5 itself while it stages them (short_kernel.hip, ASCII instantiations) and -- score-only,
    once the budgets are inherited -- the call is ONE kernel: the launch appends its failures, counts what it leaves unfinished and
    stores its partial sums straight into pinned host memory.  Same scores (and CIGARs) as the checker and as the older paths
    (tuning.no_host_parts: partial sums copied from the device; tuning.no_fused_pack: the pack kernel in front), with pairs that leave
    the tier on every way out: bytes outside ACGT (byte-compare class), a long gap (diagonal window too wide), many errors (budget),
    an empty pattern."""
    rng = random.Random(77)
    buf_a, meta_a = wfagpu.generate_pairs(24000, 150, 0.02, seed=51)
    pairs = wfagpu.pairs_from_layout(buf_a, meta_a)
    buf_b, meta_b = wfagpu.generate_pairs(60, 150, 0.25, seed=52)
    pairs += wfagpu.pairs_from_layout(buf_b, meta_b)
    for _ in range(50):
        p, t = pairs[rng.randrange(24000)]
        p = bytearray(p)
        p[rng.randrange(len(p))] = rng.choice(b"NnRYk")
        pairs.append((bytes(p), t))
    for _ in range(40):
        p, t = pairs[rng.randrange(24000)]
        cut = rng.randrange(10, 60)
        pairs.append((p, t[:cut] + t[cut + 45:]))
    pairs.append((b"", b"ACGTACGTAC"))
    pairs.append((b"ACGTTGCA", b""))
    rng.shuffle(pairs)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, PEN, cigar=cigar)
    me = int(so.max()) + 4
    # (one context: the first call draws the budget sample -- which pairs it holds is not deterministic --, the later ones inherit its
    # budgets: the one-kernel shape, then the older paths under the same budgets)
    stats = {}
    al = wfagpu.DeviceAligner(0)
    try:
        al.hint_same_stream(True)
        batch = al.upload(buf, meta)
        for name, tuning in (("sampling", {}), ("default", {}), ("parts_on_device", {"no_host_parts": 1}), ("pack_kernel", {"no_fused_pack": 1}),
                             ("default_again", {})):
            al.set_tuning(**tuning)
            s, c = al.align(batch, PEN, max_error=me, compute_cigar=cigar)
            assert np.array_equal(s, so), (name, np.nonzero(np.asarray(s) != so)[0][:8])
            if cigar:
                assert c == co, name
            st = al.stats()
            assert st.pairs_raw == 50, (name, st.pairs_raw)
            stats[name] = (st.cells, st.pairs_tier[5], st.pairs_retried, st.pairs_budget_missed, st.align_launches)
    finally:
        al.close()
    assert stats["default"][1] > 23000      # (tier 5 took the batch)
    # (the same work accounting whichever way the sums reach the host)
    assert stats["default"] == stats["parts_on_device"] == stats["pack_kernel"] == stats["default_again"], stats


@pytest.mark.parametrize("cigar", [False, True])
def test_mid_size_call_is_cut_into_batches(cigar):
    """A call that is neither big (128k+ pairs) nor tiny -- 40 000 pairs of 150 bases, 12 MB -- is cut into four batches (three with
    CIGARs) so that the upload of one runs under the record sweep of the next (wfa_launch.hip): results complete and in input order,
    equal to the checker's, pairs with bytes outside ACGT and a pair that leaves the short-read tier included."""
    import ctypes as C
    lib = wfagpu.load()
    buf_a, meta_a = wfagpu.generate_pairs(40000, 150, 0.02, seed=61)
    pairs = wfagpu.pairs_from_layout(buf_a, meta_a)
    rng = random.Random(61)
    for q in rng.sample(range(40000), 25):
        p, t = pairs[q]
        p = bytearray(p); p[rng.randrange(len(p))] = ord("N"); pairs[q] = (bytes(p), t)
    p, t = pairs[12345]
    pairs[12345] = (p, t[:30] + t[75:])
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, PEN, cigar=cigar)
    al = wfagpu.Aligner()
    assert lib.wfagpu_initialize_aligner(C.byref(al))
    try:
        for p, t in pairs:
            assert lib.wfagpu_add_sequences(C.byref(al), p, t)
        assert lib.wfagpu_initialize_parameters(C.byref(al), wfagpu.Penalties(*PEN))
        assert lib.wfagpu_set_batch_size(C.byref(al), len(pairs))      # (one batch as far as the caller goes: the CLI's default)
        al.alignment_options.max_error = int(so.max()) + 4
        al.alignment_options.compute_cigar = cigar
        for _ in range(2):
            assert lib.wfagpu_align(C.byref(al))
            n = al.num_sequence_pairs
            s = np.array([al.results[i].error for i in range(n)], dtype=np.int64)
            assert np.array_equal(s, np.asarray(so))
            if cigar:
                assert [C.string_at(al.results[i].cigar.buffer).decode() for i in range(n)] == co
            st = wfagpu.last_launch_stats()
            assert st["batches"] == (3 if cigar else 4), st
    finally:
        lib.wfagpu_destroy_aligner(C.byref(al))
        lib.wfagpu_amd_release_cache()
