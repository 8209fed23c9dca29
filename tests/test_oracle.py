"""Pins the CPU oracle (oracle/wfa_oracle.c) BEFORE it is trusted as the parity checker:
   * against the reference tests' own golden vectors (tests/golden/, see make_golden.py), and
   * against the reference's WFA2-lib compiled in place (oracle/_ref), when present.
CPU only."""
import os
import random

import numpy as np
import pytest

import oracle_lib
import wfagpu

PENS = {"p0": (1, 2, 1), "p1": (3, 1, 4), "p2": (5, 3, 2), "g231": (2, 3, 1)}


@pytest.fixture(scope="module")
def utest(golden_dir):
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "wfa.utest.seq"))
    assert len(pairs) == 305
    return pairs, wfagpu.layout_pairs(pairs)


@pytest.mark.parametrize("tag", ["p0", "p1", "p2", "g231"])
def test_oracle_matches_wfa2_utest_cigars(utest, golden_dir, tag):
    """external/WFA/tests/wfa.utest.check/test.affine.p*.alg: score AND CIGAR, 305 pairs of 5..10062 bp."""
    pairs, (buf, meta) = utest
    gs, gc = oracle_lib.read_alg(os.path.join(golden_dir, f"utest.affine.{tag}.alg"))
    s, c, _ = oracle_lib.oracle_batch(buf, meta, PENS[tag], cigar=True, nthreads=8)
    assert np.array_equal(s, gs)
    assert c == gc


@pytest.mark.parametrize("tag", ["p0", "p1", "p2"])
def test_oracle_matches_reference_score_goldens(utest, golden_dir, tag):
    """tests/data/results/test.score.affine.p*.alg of the reference (tests/test-aligner.sh:11-48)."""
    pairs, (buf, meta) = utest
    gs, _ = oracle_lib.read_alg(os.path.join(golden_dir, f"utest.score.affine.{tag}.alg"))
    s, _, _ = oracle_lib.oracle_batch(buf, meta, PENS[tag], cigar=False, nthreads=8)
    assert np.array_equal(s, gs)


@pytest.mark.parametrize("name,pens", [("seq1k", [(2, 3, 1), (5, 3, 2)]), ("seq10k", [(2, 3, 1), (3, 5, 2)])])
def test_oracle_matches_test_api_goldens(golden_dir, name, pens):
    """tests/data/sequences_1000.h / sequences_10K.h golden score arrays (tests/test_api.c:59-219)."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, f"{name}.seq"))
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen in pens:
        gold = -np.loadtxt(os.path.join(golden_dir, f"{name}.x{pen[0]}o{pen[1]}e{pen[2]}.scores"), dtype=np.int64)
        s, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False, nthreads=8)
        assert np.array_equal(s, gold[:len(s)])


def test_oracle_matches_ref_on_hifi_and_synthetic(golden_dir):
    for fname, loader in (("hifi.g231.alg", lambda: wfagpu.layout_pairs(wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq")))),
                          ("synth.cfg2.alg", lambda: wfagpu.generate_pairs(2000, 150, 0.02, 2)),
                          ("synth.cfg3.alg", lambda: wfagpu.generate_pairs(500, 1000, 0.05, 3))):
        buf, meta = loader()
        gs, gc = oracle_lib.read_alg_skip_comments(os.path.join(golden_dir, fname))
        s, c, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
        assert np.array_equal(s, gs), fname
        assert c == gc, fname


def test_oracle_matches_ref_on_long_read_goldens(golden_dir):
    """BASELINE configs[3]/[4] shapes (10 kbp @ 3 %, 30 kbp @ 10 %): goldens made with the reference's WFA2."""
    for fname, n, length, err, seed in (("synth.cfg4.alg", 64, 10000, 0.03, 44), ("synth.cfg5.alg", 16, 30000, 0.10, 55)):
        gs, gc = oracle_lib.read_alg_skip_comments(os.path.join(golden_dir, fname))
        buf, meta = wfagpu.generate_pairs(n, length, err, seed)
        s, c, _ = oracle_lib.oracle_batch(buf, meta[:len(gs)], (2, 3, 1), cigar=True, nthreads=4)
        assert np.array_equal(s, gs), fname
        assert c == gc, fname
        for (p, t), cg, sc in zip(wfagpu.pairs_from_layout(buf, meta[:len(gs)]), c, s):
            ok, cost = oracle_lib.check_cigar(p, t, cg, (2, 3, 1))
            assert ok and cost == sc


def _rand_pairs(rng, n, maxlen, alphabet=b"ACGT", err=0.1):
    out = []
    for _ in range(n):
        L = rng.randint(0, maxlen)
        t = bytes(rng.choice(alphabet) for _ in range(L))
        p = bytearray(t)
        for _ in range(int(L * err) + rng.randint(0, 2)):
            op = rng.randint(0, 2)
            if op == 0 and p:
                p[rng.randrange(len(p))] = rng.choice(alphabet)
            elif op == 1 and p:
                del p[rng.randrange(len(p))]
            else:
                p.insert(rng.randint(0, len(p)), rng.choice(alphabet))
        out.append((bytes(p), t))
    return out


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("pen", [(2, 3, 1), (1, 2, 1), (3, 1, 4), (5, 3, 2), (4, 6, 2), (1, 0, 1), (7, 2, 3)])
def test_oracle_matches_compiled_reference_random(pen):
    """Random pairs incl. empty/1-base sequences, unrelated pairs and bytes outside ACGT (WFA2 compares raw bytes)."""
    rng = random.Random(1234 + sum(pen))
    pairs = _rand_pairs(rng, 300, 60) + _rand_pairs(rng, 60, 400, err=0.25)
    pairs += _rand_pairs(rng, 40, 80, alphabet=b"ACGTNacgt")
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"A", b"A"), (b"A", b"C"), (b"ACGT", b"TGCA"),
              (b"AAAAAAAAAA", b"TTTTTTTTTTTTTTT"), (b"ACGTACGTAC", b"ACGTACGTACGTACGTACGT")]
    buf, meta = wfagpu.layout_pairs(pairs)
    rs_hi, rc_hi = oracle_lib.ref_batch(buf, meta, pen, cigar=True, memory_mode=0)
    rs_lo, rc_lo = oracle_lib.ref_batch(buf, meta, pen, cigar=True, memory_mode=1)
    s, c, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True)
    s2, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False)
    assert np.array_equal(rs_hi, rs_lo) and rc_hi == rc_lo      # WFA2 memory modes agree (utils/wfa_cpu.c uses low)
    assert np.array_equal(s, rs_hi)
    assert np.array_equal(s2, rs_hi)                             # ring (score-only) path == full path
    assert c == rc_hi
    for (p, t), cg, sc in zip(pairs, c, s):
        ok, cost = oracle_lib.check_cigar(p, t, cg, pen)
        assert ok and cost == sc


@pytest.mark.skipif(not (oracle_lib.have_ref() and oracle_lib.have_refcpu()), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("pen", [(2, 3, 1), (4, 6, 2), (5, 3, 2)])
def test_reference_cpu_functions_agree_with_the_wfa2_shim(pen):
    """utils/wfa_cpu.c itself (compute_alignments_cpu_threaded / compute_distance_cpu_threaded, compiled in place with
    utils/cigar.c: what north_star names as the CPU baseline and bench.py times as cpu_baseline.reference_shim) returns what
    the WFA2 shim (ref_shim.c) and the restatement return -- scores and CIGARs, one thread and several."""
    rng = random.Random(77 + sum(pen))
    pairs = _rand_pairs(rng, 200, 80) + _rand_pairs(rng, 40, 500, err=0.2) + [(b"A", b"A"), (b"ACGT", b"TGCA"), (b"ACGTACGTAC", b"ACGTACGTACGTACGTACGT")]
    buf, meta = wfagpu.layout_pairs(pairs)
    rs, rc = oracle_lib.ref_batch(buf, meta, pen, cigar=True, memory_mode=1)
    for nt in (1, 4):
        s, c = oracle_lib.refcpu_batch(buf, meta, pen, cigar=True, nthreads=nt)
        assert np.array_equal(s, rs) and c == rc
        s2, _ = oracle_lib.refcpu_batch(buf, meta, pen, cigar=False, nthreads=nt)
        assert np.array_equal(s2, rs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True)
    assert np.array_equal(so, rs) and co == rc


def test_host_packer_writes_the_oracle_packers_words():
    """utils/host_pack.c (launch_alignments* pack on the host when it has the cores): the vector path and the portable one
    against the oracle's packer -- every length around the 16- and 32-base steps, the zero word at the end, nothing
    written beyond it, the flag for bytes outside ACGT."""
    import ctypes as C
    lib = wfagpu.load()
    rng = random.Random(11)
    o = oracle_lib.oracle()
    lens = list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 150, 255, 256, 257, 1000, 1023, 1025] + [rng.randint(0, 3000) for _ in range(40)]
    for fn in (lib.wfagpu_host_pack_sequence, lib.wfagpu_host_pack_sequence_scalar):
        fn.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p]
        fn.restype = C.c_int
        for n in lens:
            for dirty in (False, True):
                seq = bytearray(rng.choice(b"ACGT") for _ in range(n))
                if dirty:
                    if n == 0:
                        continue
                    seq[rng.randrange(n)] = rng.choice(b"NacgtRY-\x00\xff")
                seq = bytes(seq)
                nw = (n + 15) // 16
                want = (C.c_uint32 * (nw + 1))()
                bad = o.oracle_pack2(seq, n, want)
                got = (C.c_uint32 * (nw + 3))(*([0xDEADBEEF] * (nw + 3)))
                assert fn(seq + b"GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG", n, got) == bad == (1 if dirty else 0), (n, dirty)
                if not bad:
                    assert list(got)[:nw] == list(want)[:nw], n
                assert got[nw] == 0 and got[nw + 1] == 0xDEADBEEF and got[nw + 2] == 0xDEADBEEF, n


def test_host_pack_strip_offsets_and_words():
    """wfagpu_host_pack_strip (what launch_alignments* run per strip of a batch): the offsets of
    wfagpu_amd_fill_packed_offsets, the oracle packer's words -- partial last 32 bytes through the vector path where the
    caller's buffer goes on that far, through the scalar one at its very end --, offsets only without a staging buffer, the
    flag (and no further packing) from the first byte outside ACGT on."""
    import ctypes as C
    lib = wfagpu.load()
    lib.wfagpu_host_pack_strip.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.wfagpu_host_pack_strip.restype = C.c_int
    rng = random.Random(23)
    pairs = [(bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 150, 151, 1000, rng.randint(0, 400)]))),
              bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 5, 16, 30, 32, 149, 150, rng.randint(0, 400)])))) for _ in range(300)]
    pairs.append((b"ACGTACGTACGTACGTACGTA", b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTG"))      # (the end of the buffer)
    buf, meta = wfagpu.layout_pairs(pairs)
    last = meta[-1]
    buf = np.ascontiguousarray(buf[:int(last["text_offset"]) + int(last["text_len"])])      # nothing readable after the last base
    want_meta = meta.copy()
    total = lib.wfagpu_amd_fill_packed_offsets(want_meta.ctypes.data, len(want_meta))
    o = oracle_lib.oracle()
    for first, n0 in ((0, 0), (64, 7)):
        got_meta = meta.copy()
        words = np.full(total // 4 + first // 4 + 8, 0xDEADBEEF, dtype=np.uint32)
        sub = got_meta[n0:]
        assert lib.wfagpu_host_pack_strip(buf.ctypes.data, buf.nbytes, sub.ctypes.data, len(sub), first, words.ctypes.data) == 0
        base = int(want_meta[n0]["pattern_offset_packed"]) - first
        for i in range(n0, len(pairs)):
            for seq, key in zip(pairs[i], ("pattern_offset_packed", "text_offset_packed")):
                off = int(got_meta[i][key])
                assert off == int(want_meta[i][key]) - base
                nw = (len(seq) + 15) // 16
                want = (C.c_uint32 * (nw + 1))()
                assert o.oracle_pack2(seq, len(seq), want) == 0
                assert list(words[off // 4: off // 4 + nw]) == list(want)[:nw], (i, key)
                assert words[off // 4 + nw] == 0
        assert words[(total - base) // 4] == 0xDEADBEEF if n0 == 0 else True
    only = meta.copy()
    assert lib.wfagpu_host_pack_strip(buf.ctypes.data, buf.nbytes, only.ctypes.data, len(only), 0, None) == 0
    assert np.array_equal(only["pattern_offset_packed"], want_meta["pattern_offset_packed"]) and np.array_equal(only["text_offset_packed"], want_meta["text_offset_packed"])
    dirty = bytearray(buf.tobytes())
    dirty[int(meta[100]["text_offset"]) + 3] = ord("N")
    dbuf = np.frombuffer(bytes(dirty), dtype=np.uint8)
    m3 = meta.copy()
    words = np.zeros(total // 4 + 8, dtype=np.uint32)
    assert lib.wfagpu_host_pack_strip(dbuf.ctypes.data, dbuf.nbytes, m3.ctypes.data, len(m3), 0, words.ctypes.data) == 1
    assert np.array_equal(m3["text_offset_packed"], want_meta["text_offset_packed"])      # (the offsets are assigned to the end)
    # a record that points outside the caller's buffer is flagged, not read
    m4 = meta.copy()
    m4[5]["text_offset"] = buf.nbytes + 1000
    assert lib.wfagpu_host_pack_strip(buf.ctypes.data, buf.nbytes, m4.ctypes.data, len(m4), 0, words.ctypes.data) == 1


def test_host_packer_under_address_sanitizer(tmp_path):
    """utils/host_pack.c built with -fsanitize=address,undefined (CPU build: the GPU pool has no sanitizers) against
    tests/host_pack_asan.c: records in heap buffers of exactly the bytes they need, so the vector path's 32-byte loads near
    the end of the caller's buffer and the stores around every sequence's last word land in red zones if they stray."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_pack_asan")
    cc = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(root, "include"),
                         os.path.join(root, "tests", "host_pack_asan.c"), os.path.join(root, "wfa-gpu_amd", "utils", "host_pack.c"), "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "host_pack_asan ok" in run.stdout, run.stdout + run.stderr


def test_oracle_pack_layout():
    """Bit layout of this build's packing: code=(c&6)>>1 (A0 C1 T2 G3, as tests/test_packing_kernel.cu:31 of the
    reference decodes it), base i in bits 2*(i%16) of word i//16."""
    import ctypes as C
    seq = b"ACGTTGCAACGTACGTAGGT"
    words = (C.c_uint32 * 2)()
    bad = oracle_lib.oracle().oracle_pack2(seq, len(seq), words)
    assert bad == 0
    lut = b"ACTG"
    dec = bytes(lut[(words[i // 16] >> (2 * (i % 16))) & 3] for i in range(len(seq)))
    assert dec == seq
    assert oracle_lib.oracle().oracle_pack2(b"ACGN", 4, words) == 1


def test_band_restatement_properties():
    """oracle/band_oracle.c restates the reference's adaptive-band kernel (lib/kernels/sequence_distance_kernel_aband.cu); the
    kernel itself cannot run here (CUDA), so the restatement is held by what must be true of it: with a band wider than any
    wavefront it returns the exact gap-affine score (= WFA2's) for every penalty set; identical sequences score 0; a pair whose
    end diagonal lies outside any window the band can reach within the step limit is reported unfinished; the result does not
    depend on the thread count."""
    rng = random.Random(99)
    pairs = _rand_pairs(rng, 200, 300, err=0.08) + _rand_pairs(rng, 40, 2000, err=0.05) + [(b"ACGT" * 50, b"ACGT" * 50), (b"", b"ACGT"), (b"A", b"")]
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen in ((2, 3, 1), (1, 2, 1), (3, 1, 4), (5, 3, 2), (4, 6, 2), (1, 0, 1)):
        so, _, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=False, nthreads=4)
        sb = oracle_lib.band_ref_batch(buf, meta, pen, 8192, 25, 20000, nthreads=4)
        assert np.array_equal(sb, so), pen
        assert np.array_equal(oracle_lib.band_ref_batch(buf, meta, pen, 8192, 25, 20000, nthreads=1), sb)
    same = pairs.index((b"ACGT" * 50, b"ACGT" * 50))
    assert oracle_lib.band_ref_batch(buf, meta, (2, 3, 1), 4, 10, 100)[same] == 0
    # 600 extra bases at the end of the text: the end diagonal is +600, a band of 64 that re-centres on the furthest point
    # follows it; a band that never re-centres (lambda beyond the step limit) cannot get there
    t = bytes(rng.choice(b"ACGT") for _ in range(1500))
    b2, m2 = wfagpu.layout_pairs([(t[:900], t)])
    exact = oracle_lib.oracle_batch(b2, m2, (2, 3, 1), cigar=False)[0][0]
    assert exact == 3 + 600
    assert oracle_lib.band_ref_batch(b2, m2, (2, 3, 1), 64, 10, 2000)[0] >= exact
    assert oracle_lib.band_ref_batch(b2, m2, (2, 3, 1), 64, 100000, 2000)[0] == -1
