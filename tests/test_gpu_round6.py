"""Round-6 additions, through the C-ABI on the GPU:
  * stream ordering of the device-resident entry points (EXPERIMENTS R5.8: the one red the path ever showed was the ctypes stub
    zero-filling on torch's null stream while the pack kernel ran on the context's own non-blocking stream);
  * wfagpu_amd_config_t::null_stream and a context on a torch.cuda.Stream.
The reference copies synchronously around its kernels (tests/test_packing_kernel.cu:225-306).
"""
import ctypes as C
import os
import random

import numpy as np
import pytest

import oracle_lib
import wfagpu
from test_oracle import _rand_pairs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack_expected(pairs, hm):
    """(offset in words, words incl. the spare zero, flag) per sequence, from the checker's packer."""
    o = oracle_lib.oracle()
    out = []
    for i, (p, t) in enumerate(pairs):
        for seq, off in ((p, int(hm[i]["pattern_offset_packed"])), (t, int(hm[i]["text_offset_packed"]))):
            nw = (len(seq) + 15) // 16
            words = (C.c_uint32 * (nw + 1))()
            bad = o.oracle_pack2(seq, len(seq), words)
            out.append((off // 4, np.frombuffer(words, dtype=np.uint32)[:nw].copy(), bad))
    return out


def _check_pack(packed, flags, expected):
    for j, (w0, words, bad) in enumerate(expected):
        assert flags[j] == bad, j
        if not bad:
            assert np.array_equal(packed[w0:w0 + len(words)], words), j
        assert packed[w0 + len(words)] == 0, j


@pytest.mark.parametrize("mode", ["own_stream_after_null_fill", "null_stream", "torch_stream"])
def test_pack_soak_behind_a_busy_null_stream(mode):
    """200 x (pack + compare) on a context created AFTER a large asynchronous fill was queued on the null stream, with another
    large fill queued in front of every call: the binding's own zero-fills of the output tensors sit behind it on torch's stream
    while the context packs on its stream.  Every mode must order the two (R5.8 failed once in ~10^3 runs without)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = random.Random(86)
    pairs = [(b"GATTACA", b"GATACA"), (b"ACGT" * 9, b"ACGT" * 9 + b"A"), (b"T" * 33, b"C" * 16), (b"A", b"G")]
    pairs += _rand_pairs(rng, 201, 110) + [(b"ACGNACGT", b"acgt"), (b"", b"A"), (b"", b""), (b"C" * 109, b"")]
    buf, meta = wfagpu.layout_pairs(pairs)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=dev)      # 256 MiB: a fill of ~0.1 ms and more
    big.fill_(1)
    stream = torch.cuda.Stream(dev) if mode == "torch_stream" else None
    ctx_mgr = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
    with ctx_mgr:
        al = wfagpu.DeviceAligner(0, null_stream=(mode == "null_stream"))
        try:
            if mode == "own_stream_after_null_fill":
                assert al._ctx_stream is None
            batch = al.upload(buf, meta)
            expected = _pack_expected(pairs, batch._meta_host)
            for it in range(200):
                big.fill_(it & 0x7F)
                big.fill_((it + 1) & 0x7F)
                packed, flags = al.pack(batch)
                _check_pack(packed, flags, expected)
            # and the whole path behind a busy stream: scores land in a tensor torch has just filled
            so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
            for it in range(20):
                big.fill_(it)
                d_scores = torch.full((len(pairs),), -7, dtype=torch.int32, device=dev)
                s, c = al.align(batch, (2, 3, 1), max_error=120, compute_cigar=True, d_scores=d_scores)
                assert np.array_equal(s, np.asarray(so)) and c == co
        finally:
            al.close()
    torch.cuda.synchronize()


def _truth(buf, meta, pen, cigar=True):
    if oracle_lib.have_ref():
        return oracle_lib.ref_batch(buf, meta, pen, cigar=cigar, memory_mode=0, nthreads=16)
    s, c, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=cigar, nthreads=16)
    return s, c


def test_small_batch_of_long_reads_tries_the_widest_lds_budget_first():
    """44 x 30 kbp pairs under -e 14000 (too few pairs for a budget sample; the caller's ceiling only fits the HBM-ring tier): the
    chain first runs under the largest budget the hybrid tier holds -- 40 pairs at 10 % finish there --, the 4 pairs at 16 % miss it
    and are escalated to the HBM ring.  Scores and CIGAR strings are the checker's either way (the same pairs
    took 170 ms on tier 3 against 70 on the hybrid tier)."""
    pen = (2, 3, 1)
    buf_a, meta_a = wfagpu.generate_pairs(40, 30_000, 0.10, seed=77)
    buf_b, meta_b = wfagpu.generate_pairs(4, 30_000, 0.16, seed=78)
    pairs = wfagpu.pairs_from_layout(buf_a, meta_a) + wfagpu.pairs_from_layout(buf_b, meta_b)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co = _truth(buf, meta, pen)
    assert int(so[:40].max()) < 10_000 < int(so[40:].min()) and int(so.max()) < 14_000
    al = wfagpu.DeviceAligner(0)
    try:
        batch = al.upload(buf, meta)
        for cigar in (True, False):
            s, c = al.align(batch, pen, max_error=14_000, compute_cigar=cigar)
            st = al.stats()
            assert np.array_equal(s, so)
            if cigar:
                assert c == co
            # (a pair that runs out of arena on the HBM-ring tier is launched again in the next pass and counted again)
            assert st.pairs_tier[4] == 40 and st.pairs_tier[3] >= 4 and sum(st.pairs_tier[:3]) == 0, list(st.pairs_tier)
        # a ceiling the hybrid tier holds is taken as it is
        s, c = al.align(batch, pen, max_error=9_500, compute_cigar=True)
        assert np.array_equal(s, so) and c == co
    finally:
        al.close()


def test_calls_that_skip_the_counter_memset_find_the_counters_clean():
    """A call ends with its device counters zeroed -- no kernel touched them (the one-kernel call of short reads keeps its sums in
    pinned host memory), or a memset was queued behind its last read -- and the next call starts without a memset.  That rests on every
    path that touches a counter saying so (`ct_clean`, wfa_host.hip).  tuning.verify_counters reads the block back at the start of such
    a call and fails the call if a byte is set: sequences of calls that cross the paths -- score-only on tier 5 (clean without a
    memset), with CIGARs, with pairs flagged for the byte-compare class, with budget misses, the ordinary tiers, a penalty set tier 5
    does not take -- all find it clean, and give the checker's scores."""
    rng = random.Random(77)
    buf, meta = wfagpu.generate_pairs(20000, 150, 0.02, seed=5)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    dirty = list(pairs)
    for i in range(0, len(dirty), 997):
        p, t = dirty[i]
        dirty[i] = (p[:10] + b"N" + p[11:], t)
    hard, mh = wfagpu.generate_pairs(200, 150, 0.15, seed=6)      # beyond any tuned budget: re-runs, escalations
    mixed = pairs[:19800] + wfagpu.pairs_from_layout(hard, mh)
    sets = [wfagpu.layout_pairs(x) for x in (pairs, dirty, mixed)]
    truth = {}
    al = wfagpu.DeviceAligner(0, verify_counters=1)
    try:
        batches = [al.upload(b, m) for b, m in sets]
        plan = [(0, (2, 3, 1), False), (0, (2, 3, 1), False), (0, (2, 3, 1), True), (1, (2, 3, 1), False), (0, (2, 3, 1), False),
                (2, (2, 3, 1), False), (0, (5, 3, 2), False), (0, (9, 2, 1), False), (2, (2, 3, 1), True), (0, (2, 3, 1), False), (1, (5, 3, 2), True),
                (0, (2, 3, 1), False)]
        al.hint_same_stream(True)
        for which, pen, cigar in plan:
            s, c = al.align(batches[which], pen, max_error=45 * max(1, pen[0] // 2), compute_cigar=cigar)      # (raises if the hook fails the call)
            key = (which, pen)
            if key not in truth:
                truth[key] = oracle_lib.oracle_batch(sets[which][0], sets[which][1], pen, cigar=True, nthreads=16)[:2]
            assert np.array_equal(s, truth[key][0]), (which, pen, cigar)
            if cigar:
                assert c == truth[key][1], (which, pen)
    finally:
        al.close()


def test_cli_reads_a_pipe_like_a_file(tmp_path):
    """`-i <(cat pairs.seq)` -- a process substitution: a FIFO, st_size 0 -- gives the output of `-i pairs.seq` (the mapped
    reader returned zero pairs for such inputs; the reference's getline reader takes them, utils/sequence_reader.c:137-227), and the
    tool says what it waited for in front of its wall clock."""
    import subprocess
    cli = os.path.join(ROOT, "wfa-gpu_amd", "bin", "wfa.affine.gpu")
    seq = os.path.join(ROOT, "tests", "golden", "wfa.utest.seq")
    out_file, out_pipe = str(tmp_path / "file.alg"), str(tmp_path / "pipe.alg")
    r1 = subprocess.run([cli, "-i", seq, "-x", "-o", out_file], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run(["bash", "-c", f'"{cli}" -i <(cat "{seq}") -x -o "{out_pipe}"'], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert open(out_file).read() == open(out_pipe).read() and len(open(out_file).read().splitlines()) == 305
    assert "Alignment computed. Wall time:" in r2.stdout and "Device bring-up waited for before the clock:" in r2.stdout
