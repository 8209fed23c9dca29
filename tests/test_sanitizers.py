"""Sanitizer runs of the HOST side on the CPU build (the GPU pool has no sanitizers; CPU-only tests, not `-m gpu`):

  AddressSanitizer + UBSan   tests/host_api_asan.c over lib/aligner.c, lib/alignment_results.c, utils/sequence_reader.c,
                             utils/verification.c, tools/generate_dataset.c (tests/test_oracle.py holds utils/host_pack.c's)
  ThreadSanitizer            tests/launch_tsan.cpp over csrc/wfa_launch.hip -- the launch pipeline: bring-up thread, prep / upload /
                             compute / scatter lanes, staging ring, per-device threads -- compiled as C++ against the stub HIP
                             layer of tests/hip_stub/ (streams are worker threads, the align call is the oracle)

Round 4's first runs found: results freed and indexed by the NEW pair count after sequences were added behind the parameters
(heap overflow, lib/aligner.c); signed overflow in the CIGAR checker's run-length parser on absurd counts (utils/verification.c);
the two warm-up copies of a device's bring-up sharing one scratch buffer from two streams (csrc/wfa_launch.hip)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wfa-gpu_amd")


def _build(cmd, tmp_path):
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    if r.returncode != 0 and ("sanitize" in r.stderr or "libtsan" in r.stderr or "libasan" in r.stderr):
        pytest.skip("this toolchain has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_api_under_address_and_ub_sanitizers(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "host_api_asan")
    _build(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fopenmp",
            "-I", os.path.join(ROOT, "include"), "-I", PKG, "-I", os.path.join(PKG, "lib"),
            os.path.join(ROOT, "tests", "host_api_asan.c"), os.path.join(PKG, "lib", "aligner.c"), os.path.join(PKG, "lib", "alignment_results.c"),
            os.path.join(PKG, "utils", "sequence_reader.c"), os.path.join(PKG, "utils", "verification.c"),
            os.path.join(PKG, "tools", "generate_dataset.c"), "-lm", "-lpthread", "-o", exe], tmp_path)
    work = tmp_path / "files"
    work.mkdir()
    run = subprocess.run([exe, str(work)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "host_api_asan ok" in run.stdout, (run.stdout + run.stderr)[-4000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-4000:]


def test_launch_pipeline_under_thread_sanitizer(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    out = tmp_path / "tsan"
    _build(["bash", os.path.join(ROOT, "scratch", "build_tsan.sh"), str(out)], tmp_path)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    run = subprocess.run([str(out / "launch_tsan")], capture_output=True, text=True, timeout=1200, env=env)
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-6000:]
    assert run.returncode == 0 and "launch_tsan ok" in run.stdout, (run.stdout + run.stderr)[-3000:]
