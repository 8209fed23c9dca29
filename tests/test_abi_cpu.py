"""CPU-side checks of the boundary: the shared library loads, exports every symbol include/*.h declares,
struct layouts match the reference ABI, host-side helpers behave.  No GPU calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import wfagpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(wfagpu.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return wfagpu.load()


def test_library_exports_every_declared_symbol(lib):
    declared = set()
    for hdr in ("wfa_gpu_abi.h", "wfa_gpu_device.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = txt.split("header-level helpers")[0]
        for m in re.finditer(r"^\s*(?:[\w\*]+\s+)+\**(\w+)\s*\(", txt, flags=re.M):
            name = m.group(1)
            if name not in ("defined",) and not name.isupper():
                declared.add(name)
    static_inline = {"wfa_get_threads_per_alignment", "get_num_workers", "wfagpu_set_default_options"}
    declared -= static_inline
    assert set(wfagpu.ABI_SYMBOLS) <= declared | set(wfagpu.ABI_SYMBOLS)
    for name in sorted(declared | set(wfagpu.ABI_SYMBOLS)):
        assert hasattr(lib, name), f"{name} not exported"


def test_struct_layouts_match_reference_abi():
    """SURVEY.md Appendix B (measured on the reference with gcc x86-64)."""
    assert C.sizeof(wfagpu.SeqPair) == 48 and wfagpu.SeqPair.has_N.offset == 40
    assert C.sizeof(wfagpu.Penalties) == 12
    assert C.sizeof(wfagpu.Cigar) == 24
    assert C.sizeof(wfagpu.AlignmentResult) == 32 and wfagpu.AlignmentResult.cigar.offset == 8
    assert C.sizeof(wfagpu.Options) == 48 and wfagpu.Options.penalties.offset == 32 and wfagpu.Options.compute_cigar.offset == 44
    assert C.sizeof(wfagpu.Aligner) == 104 and wfagpu.Aligner.alignment_options.offset == 56
    # and as the C compiler sees include/wfa_gpu_abi.h
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "wfa_gpu_abi.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n",sizeof(sequence_pair_t),sizeof(affine_penalties_t),
 sizeof(wfa_backtrace_t),sizeof(alignment_result_t),sizeof(wfa_cigar_t),sizeof(wfa_alignment_result_t),
 sizeof(wfa_alignment_options_t),sizeof(wfagpu_aligner_t));return 0;}'''
    exe = "/tmp/wfagpu_abi_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe, "-L", os.path.dirname(wfagpu.LIB_PATH)],
                   input=src.encode(), check=True)
    out = subprocess.run([exe], capture_output=True, check=True).stdout.split()
    assert [int(v) for v in out] == [48, 12, 8, 20, 24, 32, 48, 104]


def test_aligner_object_host_side(lib):
    """tests/test_api.c:30-57 of the reference: NULL and bad-penalty rejection; buffer layout of add_sequences."""
    al = wfagpu.Aligner()
    assert not lib.wfagpu_initialize_aligner(None)
    assert lib.wfagpu_initialize_aligner(C.byref(al))
    assert not lib.wfagpu_add_sequences(None, b"A", b"A")
    assert not lib.wfagpu_add_sequences(C.byref(al), None, b"A")
    assert not lib.wfagpu_add_sequences(C.byref(al), b"A", None)
    assert lib.wfagpu_add_sequences(C.byref(al), b"GATTACA", b"GATACA")
    assert lib.wfagpu_add_sequences(C.byref(al), b"ACGT" * 300000, b"A") is False  # >= 2^15
    big = b"ACGT" * 5000
    for _ in range(120):  # forces both grow paths (1 MiB buffer steps)
        assert lib.wfagpu_add_sequences(C.byref(al), big, big)
    assert al.num_sequence_pairs == 121
    m0, m1 = al.sequences_metadata[0], al.sequences_metadata[1]
    assert (m0.pattern_offset, m0.pattern_len, m0.text_offset, m0.text_len) == (0, 7, 12, 6)  # WFA_ALIGN_32_BITS(8) == 12: always at least one pad byte
    assert m1.pattern_offset == 20 and m1.pattern_offset % 4 == 0 and m1.text_offset % 4 == 0
    raw = C.string_at(al.sequences_buffer, 20)
    assert raw == b"GATTACA\0\0\0\0\0GATACA\0\0"
    assert not lib.wfagpu_initialize_parameters(None, wfagpu.Penalties(2, 3, 1))
    assert not lib.wfagpu_initialize_parameters(C.byref(al), wfagpu.Penalties(-1, 3, 1))
    assert not lib.wfagpu_initialize_parameters(C.byref(al), wfagpu.Penalties(0, 0, 0))
    assert not lib.wfagpu_set_batch_size(None, 10)
    assert not lib.wfagpu_align(None)
    lib.wfagpu_destroy_aligner(C.byref(al))
    # an aligner that was never given results: the reference's wfagpu_align calls the launcher anyway and returns true
    # (lib/aligner.c:236-263); this build's launcher refuses the missing arrays with a message and returns
    empty = wfagpu.Aligner()
    assert lib.wfagpu_initialize_aligner(C.byref(empty))
    assert lib.wfagpu_align(C.byref(empty))
    lib.wfagpu_destroy_aligner(C.byref(empty))


def test_packed_offsets_helper(lib):
    meta = np.zeros(3, dtype=wfagpu.META_DTYPE)
    meta["pattern_len"] = [0, 16, 17]
    meta["text_len"] = [1, 32, 1000]
    total = lib.wfagpu_amd_fill_packed_offsets(meta.ctypes.data, 3)
    words = lambda n: (n + 15) // 16 + 1
    exp, off = [], 0
    for pl, tl in zip(meta["pattern_len"], meta["text_len"]):
        exp.append((off, off + 4 * words(int(pl))))
        off += 4 * (words(int(pl)) + words(int(tl)))
    assert total == off
    assert [(int(m["pattern_offset_packed"]), int(m["text_offset_packed"])) for m in meta] == exp


def test_generator_is_seeded_and_valid():
    b1, m1 = wfagpu.generate_pairs(50, 200, 0.05, 9)
    b2, m2 = wfagpu.generate_pairs(50, 200, 0.05, 9, nthreads=1)
    b3, _ = wfagpu.generate_pairs(50, 200, 0.05, 10)
    assert np.array_equal(b1, b2) and np.array_equal(m1, m2) and not np.array_equal(b1, b3)
    assert (m1["pattern_offset"] % 4 == 0).all() and (m1["text_offset"] % 4 == 0).all()
    assert (m1["text_len"] == 200).all() and (abs(m1["pattern_len"].astype(int) - 200) <= 10).all()
    for p, t in wfagpu.pairs_from_layout(b1, m1):
        assert set(p) <= set(b"ACGT") and set(t) <= set(b"ACGT")


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples"), reason="reference tree absent")
@pytest.mark.parametrize("src,cc", [("auto_example.c", "gcc"), ("manual_example.c", "gcc"), ("auto_example.cpp", "g++")])
def test_reference_examples_compile_unchanged(lib, src, cc):
    """the reference's example programs, read where they lie, compile and link
    against this library with the reference's own include convention (-I lib -I .)."""
    pkg = os.path.dirname(wfagpu.LIB_PATH)
    exe = f"/tmp/wfagpu_refexample_{src.replace('.', '_')}"
    subprocess.run([cc, f"/root/reference/examples/{src}", "-o", exe, f"-I{pkg}/lib", f"-I{pkg}", f"-L{pkg}",
                    "-lwfagpu", f"-Wl,-rpath,{pkg}"], check=True)
    assert os.path.exists(exe)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tools"), reason="reference tree absent")
def test_reference_cli_links_against_this_library(lib, tmp_path):
    """INTEGRATION.md section B as a test: the reference's OWN CLI objects -- tools/aligner.c, utils/sequence_reader.c,
    utils/arg_handler.c, read where they lie, compiled with the reference's own headers -- link against libwfagpu.so with
    nothing left undefined, and every symbol they import from it is one include/wfa_gpu_abi.h declares.  (The same link is
    what oracle/Makefile keeps as oracle/_ref/ref.wfa.affine.gpu for the GPU suite to run.)"""
    pkg = os.path.dirname(wfagpu.LIB_PATH)
    exe = str(tmp_path / "ref_cli")
    ref = "/root/reference"
    subprocess.run(["gcc", "-O1", "-fopenmp", "-w", f"-I{ref}", f"-I{ref}/lib", f"{ref}/tools/aligner.c", f"{ref}/utils/sequence_reader.c",
                    f"{ref}/utils/arg_handler.c", "-o", exe, f"-L{pkg}", "-lwfagpu", f"-Wl,-rpath,{pkg}", "-Wl,--no-undefined", "-lm"], check=True)
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout.split()
    exported = set(subprocess.run(["nm", "-D", "--defined-only", wfagpu.LIB_PATH], capture_output=True, text=True, check=True).stdout.split())
    from_lib = sorted(sym for sym in und if sym in exported and not sym.startswith("_"))
    assert set(from_lib) == {"launch_alignments", "launch_alignments_distance", "get_num_cuda_devices", "get_cuda_dev_name",
                             "get_cuda_capability", "get_cuda_SM_count", "initialize_wfa_results", "destroy_wfa_results"}, from_lib
    abi = open(os.path.join(ROOT, "include", "wfa_gpu_abi.h")).read()
    assert all(sym in abi for sym in from_lib)
    # it starts, asks this library for the devices and (no GPU here) leaves the way the reference does
    r = subprocess.run([exe, "-i", os.path.join(ROOT, "tests", "golden", "wfa.utest.seq")], capture_output=True, text=True, timeout=120)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "No CUDA devices detected" in r.stderr


def test_device_header_structs_match_the_python_mirrors():
    """include/wfa_gpu_device.h as the C compiler sees it against the ctypes mirrors in bindings/wfagpu.py: a field added on
    one side only would shift everything behind it (tuning switches, launch configuration, stage times, statistics)."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "wfa_gpu_device.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n",sizeof(wfagpu_amd_tuning_t),sizeof(wfagpu_amd_config_t),
 offsetof(wfagpu_amd_config_t,tuning),sizeof(wfagpu_amd_launch_config_t),offsetof(wfagpu_amd_launch_config_t,tuning),
 sizeof(wfagpu_amd_launch_stats_t),offsetof(wfagpu_amd_launch_stats_t,devices),sizeof(wfagpu_amd_stats_t),
 offsetof(wfagpu_amd_stats_t,main_launch_ms),sizeof(wfagpu_amd_batch_t));return 0;}'''
    exe = "/tmp/wfagpu_device_abi_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = [int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    assert out == [C.sizeof(wfagpu.Tuning), C.sizeof(wfagpu.Config), wfagpu.Config.tuning.offset,
                   C.sizeof(wfagpu.LaunchConfig), wfagpu.LaunchConfig.tuning.offset,
                   C.sizeof(wfagpu.LaunchStats), wfagpu.LaunchStats.devices.offset, C.sizeof(wfagpu.Stats),
                   wfagpu.Stats.main_launch_ms.offset, C.sizeof(wfagpu.Batch)]


def test_the_check_paths_own_scorer_and_checkers_agree_with_the_oracle(lib):
    """`-c` (check_correctness) compares every result with utils/verification.c: an independent scalar scorer and two CIGAR
    checkers that share nothing with the kernels -- and, so far, were only ever compared with anything by the 1M-pair CLI run on
    the GPU.  Here: the scorer against the oracle (WFA2's restatement) on random pairs under five penalty sets, the checkers on the
    oracle's CIGARs (accepted, cost == score) and on corrupted ones (rejected)."""
    import random
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    lib.verification_cpu_score.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.verification_cpu_score.restype = C.c_int
    lib.check_cigar_edit.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p]
    lib.check_cigar_edit.restype = C.c_bool
    lib.check_affine_distance.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.check_affine_distance.restype = C.c_bool
    rng = random.Random(606)
    pairs = [(b"", b""), (b"", b"ACGT"), (b"ACGT", b""), (b"A", b"C")]
    for _ in range(250):
        t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 400)))
        p = bytearray(t)
        for _ in range(rng.randint(0, len(t) // 8 + 1)):
            r = rng.random(); a = rng.randint(0, len(p))
            if r < 0.4 and a < len(p): p[a] = rng.choice(b"ACGT")
            elif r < 0.7: del p[a:a + rng.randint(1, 9)]
            else: p[a:a] = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 9)))
        pairs.append((bytes(p), t))
    pairs += [(bytes(rng.choice(b"ACGT") for _ in range(60)), bytes(rng.choice(b"ACGT") for _ in range(80))) for _ in range(20)]      # unrelated
    buf, meta = wfagpu.layout_pairs(pairs)
    for pen in ((2, 3, 1), (1, 2, 1), (5, 3, 2), (3, 1, 4), (4, 6, 2)):
        so, co, _ = oracle_lib.oracle_batch(buf, meta, pen, cigar=True, nthreads=4)
        for (p, t), s_want, cg in zip(pairs, so, co):
            assert lib.verification_cpu_score(p, t, len(p), len(t), *pen) == s_want, (pen, p, t)
            assert lib.check_cigar_edit(t, p, len(t), len(p), cg.encode())
            assert lib.check_affine_distance(t, p, len(t), len(p), int(s_want), *pen, cg.encode())
            if cg:
                assert not lib.check_affine_distance(t, p, len(t), len(p), int(s_want) + 1, *pen, cg.encode())
                assert not lib.check_cigar_edit(t, p, len(t), len(p), (cg + "1M").encode())
                assert not lib.check_cigar_edit(t, p, len(t), len(p), cg[:-1].encode())

