"""GPU tests of the host-facing layers: the wfagpu_* C API (tests/test_api.c of the reference), the
launch_alignments* seam, the wfa.affine.gpu CLI (tests/test-aligner.sh, tests/test-fasta.sh) and the examples."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import wfagpu

pytestmark = pytest.mark.gpu
PKG = os.path.dirname(wfagpu.LIB_PATH)
CLI = os.path.join(PKG, "bin", "wfa.affine.gpu")


def _api_align(pairs, pen, cigar, batch=None, max_error=None):
    lib = wfagpu.load()
    al = wfagpu.Aligner()
    assert lib.wfagpu_initialize_aligner(C.byref(al))
    for p, t in pairs:
        assert lib.wfagpu_add_sequences(C.byref(al), p, t)
    assert lib.wfagpu_initialize_parameters(C.byref(al), wfagpu.Penalties(*pen))
    if batch:
        assert lib.wfagpu_set_batch_size(C.byref(al), batch)
    if max_error:
        al.alignment_options.max_error = max_error
    al.alignment_options.compute_cigar = cigar
    assert lib.wfagpu_align(C.byref(al))
    n = al.num_sequence_pairs
    scores = np.array([al.results[i].error for i in range(n)], dtype=np.int64)
    cigars = [C.string_at(al.results[i].cigar.buffer).decode() for i in range(n)] if cigar else None
    lib.wfagpu_destroy_aligner(C.byref(al))
    return scores, cigars


@pytest.mark.parametrize("batch", [None, 100, 37])
def test_api_golden_scores_1k(golden_dir, batch):
    """tests/test_api.c:167-219: 1 kbp pairs, penalties (2,3,1) and (5,3,2), several batch sizes, both modes."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))
    for pen in ((2, 3, 1), (5, 3, 2)):
        gold = -np.loadtxt(os.path.join(golden_dir, f"seq1k.x{pen[0]}o{pen[1]}e{pen[2]}.scores"), dtype=np.int64)
        s, _ = _api_align(pairs, pen, cigar=False, batch=batch)
        assert np.array_equal(s, gold[:len(s)])
        s, c = _api_align(pairs, pen, cigar=True, batch=batch)
        assert np.array_equal(s, gold[:len(s)])
        buf, meta = wfagpu.layout_pairs(pairs)
        _, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=8)
        assert c == co


def test_api_golden_scores_10k(golden_dir):
    """tests/test_api.c:59-165: 10 kbp pairs at 10 % error, (2,3,1) and (3,5,2); default max_error is 0.1*len*max(x,o,e)."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq10k.seq"))
    for pen in ((2, 3, 1), (3, 5, 2)):
        gold = -np.loadtxt(os.path.join(golden_dir, f"seq10k.x{pen[0]}o{pen[1]}e{pen[2]}.scores"), dtype=np.int64)
        s, c = _api_align(pairs, pen, cigar=True, batch=10)
        assert np.array_equal(s, gold[:len(s)])
        for (p, t), cg, sc in zip(pairs, c, s):
            ok, cost = oracle_lib.check_cigar(p, t, cg, pen)
            assert ok and cost == sc


def test_api_small_max_error_is_still_exact():
    """tests/test-aligner.sh:27 (-e 25 'test CPU recovery'): here the recovery is a wider GPU tier."""
    buf, meta = wfagpu.generate_pairs(300, 1000, 0.05, 77)
    pairs = wfagpu.pairs_from_layout(buf, meta)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    s, c = _api_align(pairs, (2, 3, 1), cigar=True, max_error=25)
    assert np.array_equal(s, so) and c == co


def _run_cli(args):
    r = subprocess.run([CLI] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


@pytest.mark.parametrize("tag,pen,e", [("p0", "1,2,1", "10000"), ("p0", "1,2,1", "25"), ("p1", "3,1,4", "10000"), ("p2", "5,3,2", "10000")])
def test_cli_score_goldens(golden_dir, tmp_path, tag, pen, e):
    """tests/test-aligner.sh of the reference, same command lines, diff against its golden files."""
    out = tmp_path / "res.out"
    r = _run_cli(["-i", os.path.join(golden_dir, "wfa.utest.seq"), "-g", pen, "-e", e, "-o", str(out)])
    assert "Alignment computed. Wall time:" in r.stdout
    got = [line.split("\t")[0] for line in open(out).read().splitlines()]
    gold = open(os.path.join(golden_dir, f"utest.score.affine.{tag}.alg")).read().split()
    assert got == gold


def test_cli_cigar_check_and_fasta(golden_dir, tmp_path):
    """tests/test-fasta.sh: paired FASTA input, -x -c, 'correct=N' on stderr; output equals the .seq run."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))
    q, t = tmp_path / "q.fasta", tmp_path / "t.fasta"
    with open(q, "w") as fq, open(t, "w") as ft:
        for i, (p, x) in enumerate(pairs):
            fq.write(f">q{i}\n")
            ft.write(f">t{i}\n")
            for j in range(0, len(p), 70):   # multi-line records
                fq.write(p[j:j + 70].decode() + "\n")
            for j in range(0, len(x), 61):
                ft.write(x[j:j + 61].decode() + "\n")
    out1, out2 = tmp_path / "a.out", tmp_path / "b.out"
    r = _run_cli(["-Q", str(q), "-T", str(t), "-x", "-c", "-b", "5", "-o", str(out1)])
    assert f"correct=5 Incorrect=0" in r.stderr and "Incorrect=1" not in r.stderr
    _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-x", "-o", str(out2)])
    assert open(out1).read() == open(out2).read() == open(os.path.join(golden_dir, "hifi.g231.alg")).read()


REF_CLI = os.path.join(os.path.dirname(PKG), "oracle", "_ref", "ref.wfa.affine.gpu")


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref/ref.wfa.affine.gpu was not built (reference tree absent at build time)")
def test_reference_cli_binary_runs_on_this_library(golden_dir, tmp_path):
    """The reference's own CLI (tools/aligner.c + utils/sequence_reader.c + utils/arg_handler.c of the reference, unmodified,
    compiled with the reference's headers by oracle/Makefile) running on libwfagpu.so: tests/test-aligner.sh's command
    lines against the reference's golden score files, -x against WFA2's score + CIGAR goldens, paired FASTA with -c (its
    own test-fasta.sh), multi-batch, and the same output as this build's CLI."""
    def ref_cli(args):
        r = subprocess.run([REF_CLI] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r
    seq = os.path.join(golden_dir, "wfa.utest.seq")
    for tag, pen in (("p0", "1,2,1"), ("p1", "3,1,4"), ("p2", "5,3,2")):
        out = tmp_path / f"ref_{tag}.out"
        r = ref_cli(["-i", seq, "-g", pen, "-e", "10000", "-b", "100", "-o", str(out)])
        assert "Alignment computed. Wall time:" in r.stdout
        assert [ln.split("\t")[0] for ln in open(out).read().splitlines()] == open(os.path.join(golden_dir, f"utest.score.affine.{tag}.alg")).read().split()
        outx, mine = tmp_path / f"ref_{tag}.x.out", tmp_path / f"mine_{tag}.x.out"
        ref_cli(["-i", seq, "-g", pen, "-e", "10000", "-x", "-o", str(outx)])
        assert open(outx).read() == open(os.path.join(golden_dir, f"utest.affine.{tag}.alg")).read()
        _run_cli(["-i", seq, "-g", pen, "-e", "10000", "-x", "-o", str(mine)])
        assert open(outx).read() == open(mine).read()
    # long reads from paired FASTA files, -x -c, batches of 5 (the reference's tests/test-fasta.sh greps the same line)
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))
    q, t = _fasta_files(tmp_path, pairs)
    # (no -o: in FASTA mode the reference's writer reads the unused .seq reader's arrays and crashes -- SURVEY.md Appendix C,
    # tools/aligner.c:497-501; its own test-fasta.sh runs without -o too.  The scores and CIGARs it was handed are checked by -c.)
    for extra in (["-x", "-c", "-b", "5"], ["-c"], ["-g", "5,2,5", "-b", "11", "-c"]):
        r = ref_cli(["-Q", str(q), "-T", str(t)] + extra)
        counts = [(int(c), int(b)) for c, b in __import__("re").findall(r"correct=(\d+) Incorrect=(\d+)", r.stderr)]
        assert counts and sum(c for c, _ in counts) == len(pairs) and all(b == 0 for _, b in counts), r.stderr[-800:]
    out2 = tmp_path / "ref_hifi.out"
    ref_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-x", "-o", str(out2)])
    assert open(out2).read() == open(os.path.join(golden_dir, "hifi.g231.alg")).read()


def test_examples_run():
    subprocess.run(["make", "-C", os.path.join(PKG, "examples")], check=True, capture_output=True)
    out = subprocess.run([os.path.join(PKG, "examples", "quickstart")], capture_output=True, text=True, check=True).stdout
    lines = out.strip().splitlines()
    assert len(lines) == 3 and lines[2].startswith("pair 2: score 0 cigar 21M")
    pairs = [(b"GATTACAGATTACAGATTACATTTGACCA", b"GATTACAGATACAGATTACATTTGGACCA"),
             (b"ACGTACGTACGTACGTTTTTACGTACGT", b"ACGTACGAACGTACGTACGTACGT")]
    for (p, t), line in zip(pairs, lines):
        s, cg, _ = oracle_lib.oracle_pair(p, t, (2, 3, 1))
        assert line.endswith(f"score {s} cigar {cg}")
    subprocess.run([os.path.join(PKG, "examples", "tuned")], check=True, capture_output=True)
    subprocess.run([os.path.join(PKG, "examples", "quickstart-cpp")], check=True, capture_output=True)


def test_cli_banded_is_valid(golden_dir, tmp_path):
    """tests/test-fasta.sh style: -B auto (re-centre every 25 scores, band = -t) on the HiFi-shaped pairs; every
    output line must be a valid alignment whose cost equals the printed score, never better than the optimum."""
    out = tmp_path / "band.out"
    _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-x", "-B", "auto", "-t", "512", "-e", "3000", "-o", str(out)])
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))
    gs, _ = oracle_lib.read_alg(os.path.join(golden_dir, "hifi.g231.alg"))
    lines = open(out).read().splitlines()
    assert len(lines) == len(pairs)
    for (p, t), line, opt in zip(pairs, lines, gs):
        sc, cg = line.split("\t")
        ok, cost = oracle_lib.check_cigar(p, t, cg, (2, 3, 1))
        assert ok and cost == -int(sc) and cost >= opt


def test_api_longest_supported_sequences():
    """lib/aligner.c:139 of the reference: sequences up to 2^15 - 1 bases; offsets then do not fit the 16-bit LDS
    tiers and the 32-bit HBM-ring tier takes the pair."""
    import random
    rng = random.Random(8)
    t = bytes(rng.choice(b"ACGT") for _ in range(32767))
    p = bytearray(t)
    for _ in range(40):
        p[rng.randrange(len(p))] = rng.choice(b"ACGT")
    del p[1000:1003]
    p = bytes(p) + b"ACG"
    assert len(p) == 32767
    pairs = [(p, t), (t[:200], t[:200]), (p[:5000], t[:5003])]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=4)
    s, c = _api_align(pairs, (2, 3, 1), cigar=True, max_error=200)
    assert np.array_equal(s, so) and c == co


def test_cached_device_state_survives_mode_and_size_changes(golden_dir):
    """launch_alignments* keep per-device state between calls: score-only and CIGAR calls of growing and
    shrinking batch sizes, with the cache dropped in between, must all give the same answers."""
    lib = wfagpu.load()
    lib.wfagpu_amd_release_cache.restype = None
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))[:240]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for step, (cigar, batch) in enumerate([(False, 30), (True, 100), (True, 7), (False, None), (True, None), (True, 64)]):
        if step == 3:
            lib.wfagpu_amd_release_cache()
        s, c = _api_align(pairs, (2, 3, 1), cigar=cigar, batch=batch)
        assert np.array_equal(s, np.asarray(so)), (step, cigar, batch)
        if cigar:
            assert c == co, (step, batch)
    lib.wfagpu_amd_release_cache()


@pytest.mark.parametrize("shards,batch", [(2, None), (3, 50), (8, 7)])
def test_call_sharded_over_several_device_slots(golden_dir, shards, batch):
    """launch_alignments* cut a call into contiguous per-device slices, one host thread +
    context + streams each, results in input order.  wfagpu_amd_launch_config_t::virtual_devices runs that path with
    several slices on the one GPU of the test box."""
    lib = wfagpu.load()
    lib.wfagpu_amd_release_cache.restype = None
    wfagpu.configure_launch(virtual_devices=shards, numa_pin=1)     # (numa_pin: slice threads move to the cores of their GPU's NUMA node, best effort)
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))[:301]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for cigar in (False, True, True):
        s, c = _api_align(pairs, (2, 3, 1), cigar=cigar, batch=batch)
        assert np.array_equal(s, np.asarray(so))
        if cigar:
            assert c == co
    wfagpu.configure_launch()
    lib.wfagpu_amd_release_cache()


# ---- reader / writer fidelity (tools/aligner.c:148-165,476-513, utils/sequence_reader.c:118-392)

def _fasta_files(tmp_path, pairs, n_q=None, n_t=None, eol="\n", width=(70, 61)):
    q, t = tmp_path / "q.fasta", tmp_path / "t.fasta"
    with open(q, "w", newline="") as fq, open(t, "w", newline="") as ft:
        for i, (p, x) in enumerate(pairs):
            if n_q is None or i < n_q:
                fq.write(f">q{i} some description{eol}")
                for j in range(0, len(p), width[0]):
                    fq.write(p[j:j + width[0]].decode() + eol)
            if n_t is None or i < n_t:
                ft.write(f">t{i}{eol}")
                for j in range(0, len(x), width[1]):
                    ft.write(x[j:j + width[1]].decode() + eol)
    return str(q), str(t)


def test_cli_num_alignments_truncates_seq_and_fasta(golden_dir, tmp_path):
    """-n (tools/aligner.c:148-165): only the first n pairs are read, in .seq and in paired FASTA input; the output has n
    lines equal to the first n golden lines; n larger than the file reads everything."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))
    gold = open(os.path.join(golden_dir, "hifi.g231.alg")).read().splitlines()
    out = tmp_path / "n.out"
    r = _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-n", "5", "-x", "-o", str(out)])
    assert open(out).read().splitlines() == gold[:5]
    assert "(5 pairs)" in r.stdout + r.stderr
    q, t = _fasta_files(tmp_path, pairs)
    _run_cli(["-Q", q, "-T", t, "-n", "3", "-x", "-o", str(out)])
    assert open(out).read().splitlines() == gold[:3]
    _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-n", "1000", "-x", "-o", str(out)])
    assert open(out).read().splitlines() == gold


def test_cli_verbose_output_and_print_to_stderr(golden_dir, tmp_path):
    """-O adds the pattern and text columns (tools/aligner.c:505-508: "%d\\t%s\\t%s\\t%s"), for .seq AND for FASTA
    input (the reference prints from the unused .seq reader there, SURVEY.md Appendix C -- fixed here); -p writes the
    same lines to stderr instead of a file; score-only mode leaves the CIGAR column empty."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))[:4]
    gold = open(os.path.join(golden_dir, "hifi.g231.alg")).read().splitlines()[:4]
    want = [f"{g}\t{p.decode()}\t{t.decode()}" for g, (p, t) in zip(gold, pairs)]
    out = tmp_path / "v.out"
    _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-n", "4", "-x", "-O", "-o", str(out)])
    assert open(out).read().splitlines() == want
    q, t = _fasta_files(tmp_path, pairs)
    _run_cli(["-Q", q, "-T", t, "-x", "-O", "-o", str(out)])
    assert open(out).read().splitlines() == want
    r = _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-n", "4", "-x", "-p"])
    printed = [ln for ln in r.stderr.splitlines() if ln[:1] == "-" or ln[:1].isdigit()]
    assert printed == gold
    r = _run_cli(["-i", os.path.join(golden_dir, "hifi.seq"), "-n", "4", "-O", "-p"])
    printed = [ln for ln in r.stderr.splitlines() if ln[:1] == "-" or ln[:1].isdigit()]
    assert printed == [f"{g.split(chr(9))[0]}\t\t{p.decode()}\t{t.decode()}" for g, (p, t) in zip(gold, pairs)]


def test_cli_unequal_fasta_record_counts_and_crlf(golden_dir, tmp_path):
    """Paired FASTA files with a different number of records align the common prefix (with a warning); CRLF line ends
    (.seq and FASTA) do not leak '\\r' into the sequences."""
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "hifi.seq"))[:6]
    gold = open(os.path.join(golden_dir, "hifi.g231.alg")).read().splitlines()[:6]
    out = tmp_path / "u.out"
    q, t = _fasta_files(tmp_path, pairs, n_q=6, n_t=4)
    r = _run_cli(["-Q", q, "-T", t, "-x", "-o", str(out)])
    assert open(out).read().splitlines() == gold[:4]
    assert "different number of records" in r.stdout + r.stderr
    q, t = _fasta_files(tmp_path, pairs, eol="\r\n")
    _run_cli(["-Q", q, "-T", t, "-x", "-c", "-o", str(out)])
    assert open(out).read().splitlines() == gold
    crlf = tmp_path / "crlf.seq"
    with open(crlf, "wb") as f:
        for p, x in pairs:
            f.write(b">" + p + b"\r\n<" + x + b"\r\n")
    _run_cli(["-i", str(crlf), "-x", "-o", str(out)])
    assert open(out).read().splitlines() == gold


def test_cli_rejects_bad_input(tmp_path):
    """Malformed .seq input and missing files end with a non-zero exit code and a message, not a crash."""
    bad = tmp_path / "bad.seq"
    bad.write_text("<ACGT\n>ACGT\n")
    r = subprocess.run([CLI, "-i", str(bad)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Malformed" in r.stdout + r.stderr
    r = subprocess.run([CLI, "-i", str(tmp_path / "nope.seq")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    r = subprocess.run([CLI], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "No input file" in r.stdout + r.stderr
    # an option the tool does not know is skipped, as by the reference's parser (utils/arg_handler.c:97-140)
    good = tmp_path / "good.seq"
    good.write_text(">ACGTACGT\n<ACGAACGT\n")
    r = subprocess.run([CLI, "-i", str(good), "--no-such-option", "-Z", "7", "-p"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "-2\t" in r.stderr


def test_cli_full_size_cfg3_with_check(tmp_path):
    """BASELINE configs[2] as the reference states it: 1M synthetic 1 kbp pairs at 5 % error, -x (CIGAR) + -c (check), through
    the CLI.  Every batch line must report Incorrect=0 (CIGAR replays onto the pair, its gap-affine cost equals the score,
    the score equals the independent CPU scorer of utils/verification.c -- lib/align.cu:258-326 of the reference), all
    pairs must be covered, and the exit code is 0 (it is 2 when any pair fails the check)."""
    import re
    seq = tmp_path / "cfg3.seq"
    n = 1_000_000
    subprocess.run([os.path.join(PKG, "bin", "generate_dataset"), "-n", str(n), "-l", "1000", "-e", "0.05", "-s", "9", "-t", "32",
                    "-o", str(seq)], check=True, timeout=600)
    out = tmp_path / "cfg3.out"
    r = subprocess.run([CLI, "-i", str(seq), "-x", "-c", "-e", "300", "-b", "250000", "-o", str(out)], capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = re.findall(r"\(Batch (\d+)\) correct=(\d+) Incorrect=(\d+)", r.stderr)
    assert lines and sum(int(c) for _, c, _ in lines) == n
    assert all(int(bad) == 0 for _, _, bad in lines), lines
    m = re.search(r"Wall time: ([0-9.]+)s", r.stdout)
    assert m
    # output file: n lines "score<TAB>CIGAR", spot-check a few against the oracle
    with open(out) as f:
        first = [next(f) for _ in range(50)]
    pairs = wfagpu.read_seq_file(str(seq), limit=50)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    assert [ln.rstrip("\n") for ln in first] == [f"{-int(s)}\t{c}" for s, c in zip(so, co)]
    os.remove(seq)
    os.remove(out)


def test_ragged_call_is_cut_by_work_not_by_count():
    """launch_alignments* cut a call into per-device slices of equal P x T work: a batch whose long pairs all sit at the end
    must still come back complete and in input order from 3 slices (virtual devices), CIGARs included."""
    import random
    lib = wfagpu.load()
    wfagpu.configure_launch(virtual_devices=3)
    rng = random.Random(77)
    from test_oracle import _rand_pairs
    pairs = _rand_pairs(rng, 600, 100, err=0.05) + _rand_pairs(rng, 40, 3000, err=0.05)
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    s, c = _api_align(pairs, (2, 3, 1), cigar=True, max_error=400)
    wfagpu.configure_launch()
    assert np.array_equal(s, np.asarray(so)) and c == co
    lib.wfagpu_amd_release_cache()


def test_budgets_inherited_across_batches_stay_exact_when_the_stream_drifts():
    """launch_alignments* reuse the auto-tuned score budgets of the first batch of a call for the later ones
    (wfagpu_amd_hint_same_stream).  Here the error rate jumps from 2 % to 10 % half-way through the call: the inherited
    budgets are wrong for every pair of batch 3, which must all be re-run (and batch 4 samples again); scores and CIGARs
    of all 36000 pairs equal the oracle's."""
    lib = wfagpu.load()
    parts = [wfagpu.generate_pairs(18000, 300, 0.02, seed=201), wfagpu.generate_pairs(18000, 300, 0.10, seed=202)]
    bufs, metas, base = [], [], 0
    for b, m in parts:
        m = m.copy(); m["pattern_offset"] += base; m["text_offset"] += base
        pad = (-len(b)) % 4
        bufs += [b, np.zeros(pad, dtype=np.uint8)]; metas.append(m); base += len(b) + pad
    buf = np.concatenate(bufs + [np.zeros(64, dtype=np.uint8)]); meta = np.concatenate(metas)
    n = len(meta)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=16)
    res = C.POINTER(wfagpu.AlignmentResult)()
    assert lib.initialize_wfa_results(C.byref(res), n, 64)
    opt = wfagpu.Options(max_error=200, threads_per_block=64, num_workers=0, band=-1, batch_size=9000, num_alignments=n,
                         penalties=wfagpu.Penalties(2, 3, 1), compute_cigar=True)
    lib.wfagpu_amd_set_num_devices(1)
    meta2 = meta.copy()
    lib.launch_alignments(buf.ctypes.data, buf.nbytes, meta2.ctypes.data, res, opt, False)
    s = np.array([res[i].error for i in range(n)], dtype=np.int64)
    c = [C.string_at(res[i].cigar.buffer).decode() for i in range(n)]
    lib.destroy_wfa_results(res, n)
    lib.wfagpu_amd_set_num_devices(0)
    assert np.array_equal(s, so)
    assert c == co


@pytest.mark.parametrize("threads,slots", [(1, 0), (3, 0), (2, 3)])
def test_launch_packs_on_the_host(golden_dir, threads, slots):
    """launch_alignments with the 2-bit packing done on the host (wfagpu_amd_launch_config_t::host_pack; automatic for big
    calls on hosts with cores to spare): same results as with the pack kernel; a batch that holds a byte outside ACGT goes
    up as ASCII (its N pairs run the byte-compare kernels), the others packed; more batches than staging buffers; the call
    sharded over several device slots."""
    lib = wfagpu.load()
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))[:290]
    pairs[130] = (pairs[130][0][:400] + b"N" + pairs[130][0][401:], pairs[130][1])
    pairs += [(b"", b"ACGT"), (b"ACGTACGTACGTACGTA", b"ACGTACGTACGTACGTT"), (b"G", b"")]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    batch = 40 if not slots else 15
    nb = (len(pairs) + batch - 1) // batch
    for mode, want_packed in ((1, nb - 1), (-1, 0), (1, nb - 1)):
        wfagpu.configure_launch(host_pack=mode, host_pack_threads=threads, virtual_devices=slots)
        for cigar in (True, False):
            s, c = _api_align(pairs, (2, 3, 1), cigar=cigar, batch=batch)
            assert np.array_equal(s, np.asarray(so))
            if cigar:
                assert c == co
            st = wfagpu.last_launch_stats()      # (of the busiest device slot)
            if not slots:
                assert st["host_packed_batches"] == want_packed and st["batches"] == nb
            else:
                assert (st["host_packed_batches"] > 0) == (mode > 0) and st["devices"] == slots
    wfagpu.configure_launch()
    lib.wfagpu_amd_release_cache()


@pytest.mark.parametrize("lanes,pool,batch", [(1, 0, 64), (3, 1 << 16, 50), (2, 1 << 16, 37), (4, 0, 25)])
def test_launch_pipeline_shapes(golden_dir, lanes, pool, batch):
    """The launch pipeline in the shapes the defaults never take on a small call: one, three and four compute lanes, and an
    input pool too small for the call (input_pool_bytes: the slots become a ring and an upload waits for the batch that
    used its slot) -- results complete, in input order, CIGARs included, twice (cached per-device state)."""
    lib = wfagpu.load()
    wfagpu.configure_launch(lanes_per_device=lanes, input_pool_bytes=pool)
    pairs = wfagpu.read_seq_file(os.path.join(golden_dir, "seq1k.seq"))[:433]
    buf, meta = wfagpu.layout_pairs(pairs)
    so, co, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=True, nthreads=8)
    for cigar in (True, False, True):
        s, c = _api_align(pairs, (2, 3, 1), cigar=cigar, batch=batch)
        assert np.array_equal(s, np.asarray(so))
        if cigar:
            assert c == co
    st = wfagpu.last_launch_stats()
    nb = (len(pairs) + batch - 1) // batch
    assert st["lanes"] == min(lanes, nb) and st["batches"] == nb
    wfagpu.configure_launch()
    lib.wfagpu_amd_release_cache()
