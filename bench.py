#!/usr/bin/env python3
"""Headline benchmark of the MI355X gap-affine WFA path (BASELINE.json metric: alignments/s + GCUPS).

One "step" = one pass of the whole hot path (2-bit pack -> wavefront kernels -> backtrace -> CIGAR text) over one
batch that is already resident in HBM.  Default workload = BASELINE.json configs[2], the configuration the north-star
target is quoted on: 1M synthetic 1 kbp pairs at 5 % error, penalties (2,3,1), score + CIGAR.  ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg4x|cfg5] [--pairs P] [--max-error E]
                  [--mode ranks|library] [--no-cpu-baseline] [--no-host-to-host]

Multi-GPU.  --mode ranks (default): one rank per GPU under torch.distributed (RCCL carries the barrier and the
max-over-ranks of the clock only; no data-path collective), every rank aligns its own P pairs: weak scaling.  When
--gpus N > 1 is given WITHOUT a torchrun environment, this script starts the N ranks itself (a `python -m
torch.distributed.run` child, before anything here touches the GPU) and relays the child's output and exit code.
--mode library: ONE process times launch_alignments() -- the reference's host-buffer call (tools/aligner.c:450-474) --
with N*P pairs sharded inside the library over N devices (what a user of the reference API gets).

Besides the contract fields the line carries
  roofline      the dominant kernel's main launch against the resource that binds it -- vector instruction issue, from the
                hardware busy counters of the committed PMC pass of the same command -- with the HBM figures (SURVEY.md 8d
                algorithmic bytes, the bytes the design moves, the bytes the counters saw) as flat hbm_* keys beside it
  cpu_baseline  the reference's WFA2 (oracle/_ref) on this box's host cores, bounded sample, N=1 only
  host_to_host  the reference's own metric: wall of launch_alignments from a pageable host buffer to host CIGAR buffers
  cli_wall      bin/wfa.affine.gpu as a fresh process on the same workload from a .seq file: the "Wall time" it prints
  configs       short legs of the other BASELINE GPU configurations (cfg2, cfg4 by policy and with the band forced, cfg5):
                value, kernel time, issue fraction, parity sample, host-to-host rate, CPU baseline each (N=1, default command)
  ont_banded    the reference's published experiment in shape: 1024 ONT-shaped 30 kbp pairs, exact against the adaptive band over
                beta x lambda: ms, speed-up, recall
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: pairs, length, error, compute_cigar, max_error, band (lambda, beta) or None, default steps, description
    "cfg2": dict(pairs=100_000, length=150, error=0.02, cigar=False, max_error=45, band=None, steps=2000,
                 desc="100k synthetic 150 bp pairs, 2% error, x=2,o=3,e=1, score-only"),
    "cfg2c": dict(pairs=100_000, length=150, error=0.02, cigar=True, max_error=45, band=None, steps=1000,
                  desc="100k synthetic 150 bp pairs, 2% error, x=2,o=3,e=1, score+CIGAR (configs[1]'s reads with CIGARs)"),
    "cfg3": dict(pairs=1_000_000, length=1000, error=0.05, cigar=True, max_error=300, band=None, steps=100,
                 desc="1M synthetic 1 kbp pairs, 5% error, x=2,o=3,e=1, score+CIGAR"),
    "cfg4": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=(25, 512), steps=200,
                 desc="16k HiFi-shaped 10 kbp pairs, 3% error, -B auto (re-centre every 25 scores) -t 512 banded, score+CIGAR"),
    "cfg4b": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=(25, 512), steps=200, force_band=True,
                  desc="16k HiFi-shaped 10 kbp pairs, 3% error, -B auto (re-centre every 25 scores) -t 512 on the BANDED kernels (tuning.force_band), score+CIGAR"),
    "cfg4x": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=None, steps=100,
                  desc="16k HiFi-shaped 10 kbp pairs, 3% error, exact (unbanded), score+CIGAR"),
    "cfg5": dict(pairs=1024, length=30_000, error=0.10, cigar=True, max_error=9000, band=None, steps=25,
                 desc="1024 ONT-shaped 30 kbp pairs, 10% error, exact (unbanded), -e 9000, score+CIGAR"),
    # the last-resort tier (the whole ring in HBM / L2: what every long pair gets whose budget no LDS tier holds -- the counterpart of the
    # reference's global ring, lib/kernels/sequence_distance_kernel.cu:214-249) on configs[4]'s pairs: tuning.min_tier = 3 (profiling leg)
    "cfg5t3": dict(pairs=1024, length=30_000, error=0.10, cigar=True, max_error=9000, band=None, steps=10, tuning={"min_tier": 3}, pairs_like="cfg5",
                   desc="1024 ONT-shaped 30 kbp pairs, 10% error, exact, -e 9000, score+CIGAR, forced onto the HBM-ring tier (tuning.min_tier = 3)"),
    # the sixteen-wave LDS tier (tier 2: what the re-runs of a long-read batch's budget misses take), configs[3]'s pairs forced onto it
    "cfg4t2": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=None, steps=20, tuning={"min_tier": 2}, pairs_like="cfg4",
                   desc="16k HiFi-shaped 10 kbp pairs, 3% error, exact, score+CIGAR, forced onto the sixteen-wave LDS tier (tuning.min_tier = 2)"),
    # the reference's kernels take any penalties (lib/kernels/sequence_distance_kernel.cu:57-160) and its own tests run (5,3,2),
    # (3,5,2) and (3,1,4) (tests/test_api.c:59-219, tests/test-aligner.sh:11-48): configs[2] / configs[1] under those sets
    "cfg3_x5o3e2": dict(pairs=1_000_000, length=1000, error=0.05, cigar=True, max_error=600, band=None, steps=50, pen=(5, 3, 2), pairs_like="cfg3",
                        desc="1M synthetic 1 kbp pairs, 5% error, x=5,o=3,e=2, score+CIGAR"),
    "cfg3_x3o1e4": dict(pairs=1_000_000, length=1000, error=0.05, cigar=True, max_error=700, band=None, steps=50, pen=(3, 1, 4), pairs_like="cfg3",
                        desc="1M synthetic 1 kbp pairs, 5% error, x=3,o=1,e=4, score+CIGAR"),
    "cfg2_x5o3e2": dict(pairs=100_000, length=150, error=0.02, cigar=False, max_error=90, band=None, steps=2000, pen=(5, 3, 2), pairs_like="cfg2",
                        desc="100k synthetic 150 bp pairs, 2% error, x=5,o=3,e=2, score-only"),
}
PEN = (2, 3, 1)      # BASELINE.json's penalties; a workload's own: wl.get("pen", PEN)


def pen_of(wl):
    return tuple(wl.get("pen", PEN))


# What one wave instruction of the hot loop's slow class takes on a SIMD (profiles/r04/valu_classes.txt) and what the lean cells cost per
# 64-diagonal chunk (align/cells_hot.inc): the design's own issue bound, cells/s = 1024 SIMDs x clock / 4.2 x 64 / 28 -- the one yardstick
# on which configs[2], [3] and [4] (and every penalty set) are comparable.
ISSUE_CYCLES_PER_VALU = 4.2
VALU_PER_CHUNK = 28
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def kernel_source_hash():
    """Identifies the kernels a PMC summary was taken with (profiles/*/..._pmc_counters.csv carry it in a '#' line)."""
    h = hashlib.sha1()
    import glob
    csrc = os.path.join(ROOT, "wfa-gpu_amd", "csrc")
    files = [os.path.join(csrc, f) for f in ("align_kernel.hip", "short_kernel.hip", "short_kernel_impl.h", "trace_kernel.hip", "pack_kernel.hip", "pack_device.h", "wfa_device.h")]
    files.append(os.path.join(ROOT, "wfa-gpu_amd", "Makefile"))      # (the compiler flags of the kernels)
    for f in files + sorted(glob.glob(os.path.join(csrc, "align", "*.inc"))):      # (the score loops and cells of align_kernel.hip)
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(seq_bytes, pairs, cells, compute_cigar):
    """SURVEY.md section 8(d), share of the wavefront kernel: packed sequences read + 48 B record + 20 B result per pair,
    + 6 B per wavefront cell (three 16-bit offsets) in CIGAR mode."""
    return int(seq_bytes + 68 * pairs + (6 * cells if compute_cigar else 0))


def _pmc_main_launch(workload, counters, optional=(), tier=None):
    """Counters of the MAIN launch of a step -- the wavefront kernel of `tier` (5: wfa_short_kernel, else wfa_align_kernel), its
    largest value per counter: the launch over the whole batch, not the sample's or a re-run's -- from the newest committed
    rocprofv3 PMC summary of this same command (profiles/rNN/<workload>_pmc_counters.csv; separate --pmc passes over one step).
    Returns (values, source, stale) -- stale when the summary was taken with other kernel sources than the ones here."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_pmc_counters.csv")))
    if not files:
        return None, None, None
    src = files[-1]
    lines = open(src).read().splitlines()
    tag = [ln for ln in lines if ln.startswith("#") and "kernel_sha1=" in ln]
    stale = (not tag) or (tag[0].split("kernel_sha1=")[1].split()[0] != kernel_source_hash())
    names = ("wfa_short_kernel", "wfa_short_score_kernel") if tier == 5 else \
            ("wfa_align_kernel",) if tier is not None else ("wfa_align_kernel", "wfa_short_kernel", "wfa_short_score_kernel")
    rows = [r for r in csv.DictReader(ln for ln in lines if not ln.startswith("#")) if any(nm in r["kernel"] for nm in names)]
    out = {}
    for c in counters:
        vals = [float(r["value"]) for r in rows if r["counter"] == c]
        if not vals:
            if c in optional:
                continue
            return None, None, None
        out[c] = max(vals)          # the main launch dominates every counter used here
        ms = [float(r["kernel_ms_under_pmc"]) for r in rows if r["counter"] == c and float(r["value"]) == out[c]]
        out["_ms_" + c] = ms[0] if ms else None
    return out, os.path.relpath(src, ROOT), stale


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(buf, meta, compute_cigar, budget_pairs, pen=PEN):
    """Times the CPU ground truth on the host cores on a bounded sample of the same workload:
    the reference's own WFA2 (oracle/_ref, kind 'reference') when it is there, else the C port."""
    import oracle_lib
    n = min(len(meta), budget_pairs)
    pairs_meta = meta[:n]
    end = int(max(pairs_meta["text_offset"].max() + pairs_meta["text_len"].max(),
                  pairs_meta["pattern_offset"].max() + pairs_meta["pattern_len"].max())) + 8
    sub = np.ascontiguousarray(buf[:end])
    cores = min(usable_cores(), 64)
    if oracle_lib.have_ref():
        kind = "reference"
        run = lambda m, nt: oracle_lib.ref_batch(sub, m, pen, cigar=compute_cigar, memory_mode=1, nthreads=nt)
    else:
        kind = "port"
        run = lambda m, nt: oracle_lib.oracle_batch(sub, m, pen, cigar=compute_cigar, nthreads=nt)
    reps = 0
    t0 = time.perf_counter()
    while True:
        run(pairs_meta, cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 2.0 or reps >= 200:
            break
    res = {"value": n * reps / dt, "unit": "alignments/s", "cores": cores, "kind": kind,
           "sample": f"{n} pairs of the same workload x{reps}, {'score+CIGAR' if compute_cigar else 'score-only'}, "
                     f"one aligner per thread, {dt:.2f} s wall"}
    # the same code on ONE core (SURVEY.md section 8d asks for both), on a sample sized for ~1-2 s
    n1 = max(1, min(n, int(n * 1.5 / max(dt / reps, 1e-6) / max(cores, 1))))
    t1 = time.perf_counter()
    run(meta[:n1], 1)
    d1 = time.perf_counter() - t1
    res["single_core"] = {"value": n1 / d1, "unit": "alignments/s", "sample": f"{n1} pairs, {d1:.2f} s wall"}
    if oracle_lib.have_refcpu():
        # the function north_star names: the reference's utils/wfa_cpu.c itself (compute_alignments_cpu_threaded /
        # compute_distance_cpu_threaded: one WFA2 aligner per OpenMP thread, memory mode low, schedule(static)), compiled in
        # place by oracle/Makefile and driven by oracle/ref_cpu_shim.c, on the same sample and the same cores
        try:
            reps2, t2 = 0, time.perf_counter()
            while True:
                oracle_lib.refcpu_batch(sub, pairs_meta, pen, cigar=compute_cigar, nthreads=cores)
                reps2 += 1
                d2 = time.perf_counter() - t2
                if d2 >= 1.0 or reps2 >= 100:
                    break
            res["reference_shim"] = {"value": n * reps2 / d2, "unit": "alignments/s", "cores": cores, "kind": "reference-shim",
                                     "what": "utils/wfa_cpu.c:%s of the reference, compiled in place (oracle/_ref/libwfacpuref.so)" %
                                             ("compute_alignments_cpu_threaded" if compute_cigar else "compute_distance_cpu_threaded"),
                                     "sample": f"{n} pairs x{reps2}, {d2:.2f} s wall"}
        except Exception as ex:
            res["reference_shim"] = {"error": str(ex)}
    return res


def host_to_host(buf, meta, wl, max_error, n_devices=1, reps=8, registered=False, tuning=None, launch_cfg=None):
    """The metric as SURVEY.md section 8(d) defines it: N / wall of launch_alignments*() -- pageable host buffers in,
    host results (CIGAR strings scattered into the caller's wfa_alignment_result_t records) out, PCIe both ways.
    First call = cold (contexts, allocations), later calls = warm (per-device state cached by the library; the arena
    cap grows after multi-pass calls, so the steady state is reached after a call or two): `warm` is the MEDIAN of the
    warm calls after the first two, `best` their minimum.  `stages_ms`: the library's own stage clock of the median-most
    call (wfagpu_amd_last_launch_stats; stages overlap)."""
    import ctypes as C
    import wfagpu
    lib = wfagpu.load()
    lib.wfagpu_amd_release_cache.restype = None
    n = len(meta)
    res = C.POINTER(wfagpu.AlignmentResult)()
    assert lib.initialize_wfa_results(C.byref(res), n, 256 if wl["cigar"] else 1)
    band = wl["band"]
    opt = wfagpu.Options(max_error=max_error, threads_per_block=band[1] if band else 64, num_workers=0,
                         band=band[0] if band else -1, batch_size=n, num_alignments=n,
                         penalties=wfagpu.Penalties(*pen_of(wl)), compute_cigar=wl["cigar"])
    fn = lib.launch_alignments if wl["cigar"] else lib.launch_alignments_distance
    wfagpu.configure_launch(num_devices=n_devices, tuning=tuning or {}, **(launch_cfg or {}))
    meta = meta.copy()

    def timed(k):
        out = []
        for _ in range(k):
            t0 = time.perf_counter()
            fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
            out.append(((time.perf_counter() - t0) * 1e3, wfagpu.last_launch_stats()))
        return out

    lib.wfagpu_amd_release_cache()
    # (what an earlier leg of this process held goes back to the driver here, and the driver wipes released memory before it hands it
    # out again: a large allocation right behind a large release waits for that -- a cold call of 511 ms instead of 27-34 seen once
    # for cfg4 behind the headline's 60 GB.  A cold call of a fresh process has nothing to wait for: give the wipe a moment.)
    time.sleep(0.3)
    calls = timed(3 + reps)
    ms = [c[0] for c in calls]
    steady = sorted(calls[3:], key=lambda c: c[0])
    med_ms, med_stats = steady[len(steady) // 2]
    out = {"unit": "alignments/s", "pairs": n, "devices": n_devices,
           "what": "wall of launch_alignments%s(): pageable host buffer -> device -> host results%s" %
                   ("" if wl["cigar"] else "_distance", " with CIGAR strings" if wl["cigar"] else ""),
           "pageable": {"cold_ms": round(ms[0], 2), "warm_ms": round(med_ms, 2), "best_ms": round(steady[0][0], 2),
                        "cold": round(n / ms[0] * 1e3, 1), "warm": round(n / med_ms * 1e3, 1),
                        "best": round(n / steady[0][0] * 1e3, 1), "calls_ms": [round(m, 2) for m in ms],
                        # no call after the cold one may take a multiple of the median (an arena that is freed and re-allocated
                        # between calls stalls behind the driver's wipe of released memory: profiles/r04/cold_long.txt)
                        "max_over_median": round(max(ms[1:]) / sorted(ms[1:])[len(ms[1:]) // 2], 3)},
           "stages_ms": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in med_stats.items()},
           "cold_stages_ms": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in calls[0][1].items()},
           "input_bytes": int(buf.nbytes)}
    if registered:
        hip = wfagpu._hiprt()
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        hip.hipHostUnregister.argtypes = [C.c_void_p]
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(buf.ctypes.data, buf.nbytes, 0)
        reg_ms = (time.perf_counter() - t0) * 1e3
        if rc == 0:
            ms = [c[0] for c in timed(reps)]
            hip.hipHostUnregister(buf.ctypes.data)
            out["registered"] = {"register_ms": round(reg_ms, 2), "warm_ms": round(min(ms), 2),
                                 "warm": round(n / min(ms) * 1e3, 1)}
    out["first_result"] = int(res[0].error)
    lib.destroy_wfa_results(res, n)
    wfagpu.configure_launch()
    lib.wfagpu_amd_release_cache()      # (lanes, arenas, input slots: back to the driver before the next leg)
    return out


def self_spawn(args):
    """--gpus N without a torchrun environment: start the N ranks as a CHILD of this process -- nothing here has touched
    the GPU yet (no HIP call, no torch.cuda.*), and the parent never execs: it relays output and exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def build_roofline(workload, st, main_ms, launches_per_step, all_ms, compute_cigar, use_pmc):
    """`roofline` of the dominant kernel (wfa_align_kernel / wfa_short_kernel), MAIN launch of a step.  Flat on
    purpose (scalars only).  What binds these kernels is vector instruction issue, so `frac` is the VALU-busy fraction:
    SQ_ACTIVE_INST_VALU (quad-cycles the vector pipe spent on instructions, summed over the chip) of the committed PMC pass
    of this same command, x 4 cycles / 1024 SIMDs, over the LIVE launch duration at the clock the PMC pass ran at.  The HBM
    figures sit beside it: hbm_algorithmic_* is SURVEY.md 8(d) (6 B per cell = three 16-bit offsets), hbm_design_* what this
    design moves (1 origin byte per cell + 8 B per score row + packed sequences + records/results), hbm_counter_* what the
    TCC counters saw; without a PMC pass for the workload the line falls back to bound = "hbm" on the algorithmic bytes."""
    pairs, cells, seqb = int(st.main_launch_pairs), int(st.main_launch_cells), int(st.main_launch_seq_bytes)
    alg = algorithmic_bytes(seqb, pairs, cells, compute_cigar)
    # what the design moves: packed sequences + record + result per pair and, with CIGARs, the backtrace arena it fills -- one
    # origin byte per cell in 16-byte units + 8 B per score of row table (arena_units: of the call, i.e. the main launch + its re-runs)
    design = int(seqb + 68 * pairs + (16 * int(st.arena_units) if compute_cigar else 0))
    secs = main_ms * 1e-3
    hw = ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"]
    pmc, pmc_src, pmc_stale = _pmc_main_launch(workload, ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"] + hw, optional=hw,
                                               tier=int(st.main_launch_tier)) if use_pmc else (None, None, None)
    # HBM bytes of the main launch: (FETCH_SIZE*2 + WRITE_SIZE) KB -- FETCH_SIZE counts half of a wide coalesced read on
    # gfx950 (MI355X_MICROARCH.md, HBM section)
    traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024) if pmc else None
    r = {"kernel": "wfa_short_kernel (main launch of a step)" if int(st.main_launch_tier) == 5 else "wfa_align_kernel (main launch of a step)",
         "kernel_ms": round(main_ms, 4), "tier": int(st.main_launch_tier), "pairs_per_launch": pairs, "cells_per_launch": cells,
         "cells_per_s": round(cells / secs, 1) if secs > 0 else None,
         "launches_per_step": launches_per_step, "all_launches_ms_per_step": round(all_ms, 4), "cells_per_step": int(st.cells),
         "traffic": traffic, "traffic_source": pmc_src, "pmc_stale": pmc_stale,
         "hbm_peak_GBps": HBM_PEAK_GBS,
         "hbm_algorithmic_bytes": alg, "hbm_algorithmic_GBps": round(alg / secs / 1e9, 2) if secs > 0 else None,
         "hbm_algorithmic_frac": round(alg / secs / 1e9 / HBM_PEAK_GBS, 5) if secs > 0 else None,
         "hbm_design_bytes": design, "hbm_counter_bytes": traffic,
         "hbm_counter_GBps": round(traffic / secs / 1e9, 2) if traffic and secs > 0 else None,
         "hbm_frac_of_peak": round(traffic / secs / 1e9 / HBM_PEAK_GBS, 5) if traffic and secs > 0 else None}
    if pmc and secs > 0 and pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("GRBM_GUI_ACTIVE") and pmc.get("_ms_GRBM_GUI_ACTIVE"):
        simds, xcds = 1024, 8
        cyc = pmc["GRBM_GUI_ACTIVE"] / xcds                                  # shader cycles of the launch under the PMC pass
        clock = cyc / (pmc["_ms_GRBM_GUI_ACTIVE"] * 1e-3)                    # Hz during that pass
        busy = pmc["SQ_ACTIVE_INST_VALU"] * 4.0                              # VALU-busy SIMD-cycles of the launch (work-invariant)
        peak = simds * clock / 1e9                                           # G SIMD-cycles per second the chip has
        ach = busy / secs / 1e9
        # `frac`: VALU busy INSIDE the PMC pass (busy SIMD-cycles / SIMD-cycles of that pass: a measurement); the same busy cycles over
        # the faster un-profiled launch at the PMC pass's clock (`frac_live_upper_bound`) is an upper bound, not a measurement.
        busy_pmc = busy / simds / cyc
        issue_peak = simds * clock / ISSUE_CYCLES_PER_VALU * 64.0 / VALU_PER_CHUNK          # cells/s of the design's hot loop at full issue
        r.update({"bound": "valu-issue", "unit": "G VALU-busy SIMD-cycles/s", "achieved": round(busy_pmc * peak, 2), "peak": round(peak, 2),
                  "frac": round(busy_pmc, 4),
                  "frac_formula": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), both of the committed PMC pass",
                  "frac_live_upper_bound": round(ach / peak, 4),
                  "issue_bound_cells_per_s": round(issue_peak, 1),
                  "issue_bound_frac": round(cells / secs / issue_peak, 4),
                  "issue_bound_formula": "cells/s / (1024 SIMDs x clock / 4.2 cycles per wave instruction x 64 cells / 28 instructions per chunk)",
                  "clock_ghz_pmc_pass": round(clock / 1e9, 3), "pmc_kernel_ms": round(pmc["_ms_GRBM_GUI_ACTIVE"], 4),
                  "valu_busy_pmc_pass": round(busy / simds / cyc, 4),
                  "salu_busy_pmc_pass": round(pmc["SQ_ACTIVE_INST_SCA"] * 4 / simds / cyc, 4) if pmc.get("SQ_ACTIVE_INST_SCA") else None,
                  "lds_busy_pmc_pass": round(pmc["SQ_ACTIVE_INST_LDS"] * 4 / simds / cyc, 4) if pmc.get("SQ_ACTIVE_INST_LDS") else None,
                  "valu_insts_per_launch": int(pmc["SQ_INSTS_VALU"]), "salu_insts_per_launch": int(pmc["SQ_INSTS_SALU"]),
                  "valu_insts_per_cell": round(pmc["SQ_INSTS_VALU"] * 64 / max(cells, 1), 3)})
        if pmc.get("SQ_WAVE_CYCLES"):
            for k, c in (("wave_cycles_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_cycles_stalled_at_issue", "SQ_WAIT_INST_ANY"),
                         ("wave_cycles_parked_waitcnt_or_barrier", "SQ_WAIT_ANY")):
                if c in pmc:
                    r[k] = round(pmc[c] / pmc["SQ_WAVE_CYCLES"], 3)
    else:
        # (no counters for this command: the issue bound at the chip's nominal engine clock)
        nominal = 1024 * 2.4e9 / ISSUE_CYCLES_PER_VALU * 64.0 / VALU_PER_CHUNK
        r.update({"issue_bound_cells_per_s": round(nominal, 1), "issue_bound_frac": round(cells / secs / nominal, 4) if secs > 0 else None,
                  "issue_bound_formula": "cells/s / (1024 SIMDs x 2.4 GHz nominal / 4.2 cycles per wave instruction x 64 cells / 28 instructions per chunk)"})
        r.update({"bound": "hbm", "unit": "GB/s", "achieved": r["hbm_algorithmic_GBps"], "peak": HBM_PEAK_GBS, "frac": r["hbm_algorithmic_frac"],
                  "frac_formula": "SURVEY 8(d) algorithmic bytes / live kernel seconds / 8 TB/s (no PMC pass committed for this command: "
                                  "the kernel is issue-bound, see DESIGN.md)"})
    return r


def parity_sample(buf, meta, wl, n_pairs, d_scores, ptrs, st, band):
    """Parity spot check OUTSIDE the timed region: a sample of the last timed step's outputs against the checker."""
    import oracle_lib
    import wfagpu
    PEN = pen_of(wl)      # (this function's penalties)
    try:
        k = min(2000 if wl["length"] <= 1000 else (64 if wl["length"] <= 10000 else 8), n_pairs)
        if band and wl.get("force_band"):
            k = min(1024, n_pairs)      # (the banded-CIGAR corrections are counted on a larger sample: scores against the band rule and WFA2 scores only)
        scores = d_scores[:k].cpu().numpy()
        if oracle_lib.have_ref():
            so, co = oracle_lib.ref_batch(buf, meta[:k], PEN, cigar=wl["cigar"], memory_mode=0, nthreads=min(16, usable_cores()))
        else:
            so, co, _ = oracle_lib.oracle_batch(buf, meta[:k], PEN, cigar=wl["cigar"], nthreads=min(8, usable_cores()))
        cg = wfagpu.fetch_cigars(ptrs[0], ptrs[1], ptrs[2], n_pairs, st.text_bytes)[:k] if wl["cigar"] else None
        if band:
            # the adaptive band is a heuristic: valid alignments, cost == score >= optimum, and never worse than the
            # reference's own band rule restated on the CPU (oracle/band_oracle.c) gives for the pair; recall reported
            pairs = wfagpu.pairs_from_layout(buf, meta[:k])
            chk = [oracle_lib.check_cigar(p, t, c, PEN) for (p, t), c in zip(pairs, cg)]
            ok = all(o and cost == s for (o, cost), s in zip(chk, scores)) and bool((scores >= so).all())
            sr = oracle_lib.band_ref_batch(buf, meta[:k], PEN, band[1], band[0], wl["max_error"], nthreads=min(16, usable_cores()))
            want = np.where(sr >= 0, sr, so)
            return {"pairs": k, "valid_and_cost_equals_score": ok, "recall": float((scores == so).mean()),
                    "not_above_the_reference_band_rule": bool((scores <= want).all()),
                    # (CIGAR mode reports the cost of the CIGAR returned: below the rule's forward score where two gaps of a kind
                    # that the band made the search open back to back print as one)
                    "pairs_below_the_rules_forward_score": int((scores < want).sum()),
                    # (each such place saves whole gap openings: the reported cost below the rule's forward score, min .. max)
                    "saved_below_the_rule_min_max": [int((want - scores)[scores < want].min()), int((want - scores)[scores < want].max())] if (scores < want).any() else [0, 0],
                    "equal_to_the_reference_band_rule": int((scores == want).sum()),
                    "reference_band_rule_recall": float(((sr >= 0) & (sr == so)).mean())}
        return {"pairs": k, "bit_exact_vs_oracle": bool(np.array_equal(scores, so)) and (not wl["cigar"] or cg == co)}
    except Exception as ex:  # the checker is optional for the measurement itself
        return {"error": str(ex)}


def make_pairs(name, n_pairs, seed, world=1):
    """(world: ranks generating at once on this host -- each takes its share of the cores, not 16 threads apiece)"""
    import wfagpu
    wl = WORKLOADS[name]
    return wfagpu.generate_pairs(n_pairs, wl["length"], wl["error"], seed=seed, nthreads=max(1, min(16, usable_cores() // max(1, world))))


def settle(seconds=3.0):
    """Device memory that a process (or an earlier leg of this one) has released is wiped by the kernel driver in the
    background before it can be handed out again (KFD: wipe on release, ~20 GB/s); a large hipMalloc that follows a large free
    stalls behind that -- seen as single hipMalloc calls of 1.3-2.9 s in the FIRST launch_alignments call after a 20-60 GB free
    (profiles/r04/cold_long.txt), which is what round 3's driver run billed to the cold call.  The cold legs are measured on
    a device that has been left alone for a moment, like a freshly started CLI would find it."""
    time.sleep(seconds)


def resident_leg(name, n_pairs, max_error, steps, warmup, device, seed, tuning, inherit_budgets, dist=None, use_pmc=True, data=None):
    """One workload, resident in HBM: W warm-up steps, K timed steps (barrier + device sync on both sides, max over ranks).
    Returns (record, buf, meta): the measurements of this rank's run (+ the parity sample, taken from the last timed step)."""
    import torch
    import shardlib
    import wfagpu
    wl = WORKLOADS[name]
    buf, meta = data if data is not None else make_pairs(name, n_pairs, seed)
    al = wfagpu.DeviceAligner(device, **tuning)
    batch = al.upload(buf, meta)
    dptt = int((meta["pattern_len"].astype(np.int64) * meta["text_len"].astype(np.int64)).sum())
    acc = {"align_ms": 0.0, "pack_ms": 0.0, "trace_ms": 0.0, "launches": 0, "main_ms": 0.0}
    last = {}
    band = wl["band"]

    # (the scores of every step go into ONE device buffer, like the results array of a caller of launch_alignments*)
    d_scores_buf = torch.empty(max(len(meta), 1), dtype=torch.int32, device=torch.device("cuda", device))

    def step():
        last["out"] = al.align(batch, pen_of(wl), max_error=max_error, compute_cigar=wl["cigar"], band=band[0] if band else -1,
                               band_width=band[1] if band else 0, fetch=False, d_scores=d_scores_buf)
        st = al.stats()
        acc["align_ms"] += st.align_ms
        acc["pack_ms"] += st.pack_ms
        acc["trace_ms"] += st.trace_ms
        acc["launches"] += st.align_launches
        acc["main_ms"] += st.main_launch_ms

    for _ in range(warmup):
        step()
    # The steps of a run are batches of one stream of reads: like launch_alignments* does for the batches of a call, the
    # score budgets tuned on a sample during the warm-up are tried again without sampling (results stay exact: a pair that
    # misses its budget is re-run, and the parity sample is taken from the last timed step).
    inherit = warmup > 0 and inherit_budgets
    if inherit:
        al.hint_same_stream(True)
    for k in acc:
        acc[k] = 0
    elapsed, per_rank = shardlib.timed_steps(step, steps, 0, dist=dist, sync=torch.cuda.synchronize, device="cuda", per_rank=True)
    d_scores, ptrs = last["out"]
    st = al.stats()
    main_ms = acc["main_ms"] / steps
    rec = {"elapsed": elapsed, "per_rank_elapsed": per_rank, "dptt": dptt, "inherit": inherit, "n_pairs": n_pairs, "steps": steps,
           "ms_per_step": elapsed / steps * 1e3,
           "stage_ms_per_step": {"pack": round(acc["pack_ms"] / steps, 3), "align": round(acc["align_ms"] / steps, 3),
                                 "trace": round(acc["trace_ms"] / steps, 3)},
           "tiers": {"lds_bytes_first": int(st.lds_bytes_tier0), "blocks_per_cu_first": int(st.blocks_per_cu_tier0), "waves_per_simd_first": int(st.waves_per_simd_tier0),
                     "pairs_per_tier": [int(v) for v in st.pairs_tier], "pairs_retried": int(st.pairs_retried),
                     "pairs_banded": int(st.pairs_banded), "auto_budget": int(st.auto_budget),
                     "pairs_budget_missed": int(st.pairs_budget_missed), "passes": int(st.sub_batches),
                     "arena_gb": round(st.arena_units * 16 / 1e9, 2)},
           "roofline": build_roofline(name, st, main_ms, acc["launches"] / steps, acc["align_ms"] / steps, wl["cigar"], use_pmc),
           "parity_sample": parity_sample(buf, meta, wl, n_pairs, d_scores, ptrs, st, band)}
    al.close()
    del batch
    torch.cuda.empty_cache()
    torch.cuda.synchronize()      # (the resident leg's device memory is back with the driver before a host-to-host leg starts)
    return rec, buf, meta


def cli_wall(n_pairs=1_000_000, length=1000, error=0.05, max_error=300, runs=2):
    """The CLI as a user runs it: bin/wfa.affine.gpu -i <file> -x in a FRESH process per run (every CLI invocation is a cold
    launch_alignments call).  The number is the "Wall time" the tool prints (tools/aligner.c:450-474 of the reference: the
    launch_alignments call alone -- file reading and output writing are outside it)."""
    import re
    import tempfile
    pkg = os.path.join(ROOT, "wfa-gpu_amd")
    cli, gen = os.path.join(pkg, "bin", "wfa.affine.gpu"), os.path.join(pkg, "bin", "generate_dataset")
    if not (os.path.exists(cli) and os.path.exists(gen)):
        return {"error": "bin/wfa.affine.gpu or bin/generate_dataset not built"}
    with tempfile.TemporaryDirectory(prefix="wfagpu_cli_") as tmp:
        seq = os.path.join(tmp, "cfg3.seq")
        t0 = time.perf_counter()
        subprocess.run([gen, "-n", str(n_pairs), "-l", str(length), "-e", str(error), "-s", "9", "-t", str(min(16, usable_cores())), "-o", seq],
                       check=True, timeout=600, capture_output=True)
        gen_s = time.perf_counter() - t0
        # One untimed run on a small file first: on a box that has just been handed out the tool's shared objects (the HIP runtime,
        # its code-object compiler, this library) still come off the image's disk, and the first process that maps them waits for
        # that -- 381 ms and 129 ms for the first two timed runs on such a box against 60-70 ms on one that has run anything (the
        # pattern of BENCH_r04's 321 / 120 ms).  The timed runs below are fresh processes all the same.
        small = os.path.join(tmp, "small.seq")
        subprocess.run([gen, "-n", "2000", "-l", str(length), "-e", str(error), "-s", "8", "-o", small], check=True, timeout=600, capture_output=True)
        t0 = time.perf_counter()
        subprocess.run([cli, "-i", small, "-x", "-e", str(max_error)], capture_output=True, text=True, timeout=600)
        first_process_s = round(time.perf_counter() - t0, 3)
        walls, proc, stages, reads, waits = [], [], [], [], []
        for _ in range(runs):
            t0 = time.perf_counter()
            # (--stage-times: the library's own stage clocks of the call on stderr, so that a slow run says where it was slow)
            r = subprocess.run([cli, "-i", seq, "-x", "-e", str(max_error), "--stage-times"], capture_output=True, text=True, timeout=600)
            proc.append(round(time.perf_counter() - t0, 3))
            m = re.search(r"Wall time: ([0-9.]+)s \(([0-9.]+) alignments per second\)", r.stdout)
            bw = re.search(r"Device bring-up waited for before the clock: ([0-9.]+)s", r.stdout)
            waits.append(float(bw.group(1)) if bw else None)
            if r.returncode != 0 or not m:
                return {"error": f"exit code {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
            walls.append((float(m.group(1)), float(m.group(2))))
            st = re.search(r"\[wfagpu timing\] device 0 [^\n]*", r.stderr)
            bu = re.search(r"\[wfagpu timing\]   bring-up: [^\n]*", r.stderr)
            rd = re.search(r"File read: ([0-9.]+)s", r.stderr)
            cs = re.search(r"\[cli stages\] [^\n]*", r.stderr)      # (the tool's own stages: read, results array, call, output, release)
            stages.append(((st.group(0)[len("[wfagpu timing] "):] if st else "") + (" | " + bu.group(0)[len("[wfagpu timing]   "):] if bu else "") +
                           (" | " + cs.group(0)[1:] if cs else "")) or None)
            reads.append(float(rd.group(1)) if rd else None)
    best = min(walls)
    return {"what": "bin/wfa.affine.gpu -i <1M x 1 kbp @ 5 % .seq> -x -e 300, fresh process per run: the 'Wall time' line it prints",
            "unit": "alignments/s", "pairs": n_pairs, "value": round(max(w[1] for w in walls), 1), "wall_s": [w[0] for w in walls],
            "alignments_per_s": [w[1] for w in walls], "best_wall_ms": round(best[0] * 1e3, 1), "process_s": proc, "file_read_s": reads,
            # (the tool waits for the device's background bring-up in FRONT of its clock -- the reference's CUDA context exists when its
            # clock starts, its allocations and module load do not: the wait is reported beside the wall it is not part of)
            "bring_up_wait_s": waits, "bring_up_wait_ms": round(min(w for w in waits if w is not None) * 1e3, 1) if any(w is not None for w in waits) else None,
            "wall_plus_bring_up_wait_ms": round(min((w[0] + (bw or 0.0)) for w, bw in zip(walls, waits)) * 1e3, 1),
            "stage_clocks": stages, "generate_s": round(gen_s, 2),
            "untimed_first_process_s": first_process_s, "untimed_first_process": "the tool on 2000 pairs, once, before the timed runs: maps the shared objects on a box that may not have run anything yet"}


ONT_POINTS = ((512, 25), (1024, 25), (1024, 100))      # (beta, lambda) kept in the default line; --ont-banded-only runs the whole grid


def ont_banded_leg(n=1024, length=30_000, reps=2, full=False):
    """The reference's only published experiment, in shape (README.md:125-137 of the reference, img/approximate-time.png /
    approximate-recall.png; BASELINE.md section 1: exact ~4400 s against beta 512 ~990-1900 s and beta 1024 ~1640-5070 s, recall
    96.8-99.9 %, on a Nanopore set): ONT-shaped 30 kbp pairs, the exact search against the adaptive band over the same grid,
    beta in {352, 512, 1024} x lambda in {10, 25, 50, 100, 750}: step ms (resident batch, score + CIGAR), speed-up over the exact
    step, recall (share of pairs with the optimal score) and the pairs that finished inside the band.  Two data sets: the i.i.d.
    model of BASELINE configs[4] (10 % single-base edits) and a long-read shaped one (6 % events of which 60 % are indels of
    2.5 bases on average, 2 % long ones of 30-150 bases, 30 % clustered: ~10 % of the bases differ)."""
    import torch
    import wfagpu
    nt = min(16, usable_cores())
    sets = (("iid_10pct", lambda: wfagpu.generate_pairs(n, length, 0.10, seed=1000, nthreads=nt), 9000),
            ("long_read_shaped", lambda: wfagpu.generate_pairs_model(n, length, seed=1001, error=0.06, indel_frac=0.6, indel_mean=2.5, long_frac=0.02,
                                                                     long_min=30, long_max=150, cluster=0.3, nthreads=nt), 12000))
    out = {"what": f"{n} x {length // 1000} kbp pairs, penalties x=2,o=3,e=1, score + CIGAR, resident batch; ms = best of {reps} steps after a warm-up",
           "grid": "beta x lambda = the reference's README.md:125-137" if full else
                   "three operating points of the reference's beta x lambda grid (README.md:125-137); the whole grid: `bench.py --ont-banded-only`, "
                   "committed as profiles/r06/ont_banded_grid.json"}
    for name, gen, me in sets:
        buf, meta = gen()
        al = wfagpu.DeviceAligner(0)
        batch = al.upload(buf, meta)

        def run(band, beta):
            best = None
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d_scores, _ = al.align(batch, PEN, max_error=me, compute_cigar=True, band=band, band_width=beta, fetch=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                st = al.stats()
                if best is None or dt < best[0]:
                    best = (dt, float(st.align_ms), int(st.pairs_banded), int(st.main_launch_tier))
            return best, d_scores.cpu().numpy()

        ex, s_exact = run(-1, 0)
        rec = {"max_error": me, "mean_score": round(float(s_exact.mean()), 1), "exact_ms": round(ex[0], 2), "exact_kernel_ms": round(ex[1], 2), "exact_tier": ex[3],
               "rows": []}
        for beta in (352, 512, 1024):
            for lam in (10, 25, 50, 100, 750):
                if not full and (beta, lam) not in ONT_POINTS:
                    continue
                r, sc = run(lam, beta)
                rec["rows"].append({"beta": beta, "lambda": lam, "ms": round(r[0], 2), "kernel_ms": round(r[1], 2), "tier": r[3],
                                    "speedup": round(ex[0] / r[0], 2), "kernel_speedup": round(ex[1] / r[1], 2),
                                    "recall": round(float((sc == s_exact).mean()), 4), "inside_band": r[2],
                                    "mean_excess": round(float(((sc - s_exact) / np.maximum(s_exact, 1)).mean()), 5)})
        out[name] = rec
        al.close()
        del batch
        torch.cuda.empty_cache()
    return out


def extra_config(name, steps, warmup, data=None, h2h_leg=True, cpu_leg=True):
    """A short leg of another configuration on this GPU, everything the headline carries in small: resident value,
    kernel time + issue fraction + cells/s, parity sample, host-to-host rate, CPU baseline.  data: the pairs of a workload that shares
    another's (configs[2] under other penalties: the headline's own pairs)."""
    wl = WORKLOADS[name]
    force_band = bool(wl.get("force_band"))
    tuning = {"force_band": 1} if force_band else {}
    tuning.update(wl.get("tuning", {}))
    t0 = time.perf_counter()
    n = wl["pairs"]
    buf, meta = data if data is not None else make_pairs(wl.get("pairs_like", name), n, 1000)
    h2h_out = {}
    if h2h_leg:
        # (host to host FIRST, on a device nobody has just released tens of GB on: see settle())
        settle(2.0)
        try:
            h2h = host_to_host(buf, meta, wl, wl["max_error"], tuning=tuning, reps=3)
            h2h_out = {"host_to_host_value": h2h["pageable"]["warm"], "host_to_host_ms_per_call": h2h["pageable"]["warm_ms"],
                       "host_to_host_cold_ms": h2h["pageable"]["cold_ms"], "host_to_host_calls_ms": h2h["pageable"]["calls_ms"],
                       "host_to_host_max_over_median": h2h["pageable"]["max_over_median"]}
        except Exception as ex:
            h2h_out = {"host_to_host_value": None, "host_to_host_error": str(ex)}
    rec, buf, meta = resident_leg(name, n, wl["max_error"], steps, warmup, 0, 1000, tuning, True, use_pmc=True, data=(buf, meta))
    rf = rec["roofline"]
    out = {"workload": wl["desc"], "pairs": n, "steps": steps, "warmup": warmup, "penalties": "x=%d,o=%d,e=%d" % pen_of(wl),
           "value": round(n * steps / rec["elapsed"], 1), "unit": "alignments/s", "ms_per_step": round(rec["ms_per_step"], 3),
           "gcups": round(rec["dptt"] * steps / rec["elapsed"] / 1e9, 2),
           "kernel_ms": rf["kernel_ms"], "tier": rf["tier"], "cells_per_launch": rf["cells_per_launch"], "cells_per_s": rf["cells_per_s"],
           "issue_bound_frac": rf.get("issue_bound_frac"), "roofline_bound": rf["bound"], "roofline_frac": rf["frac"],
           "roofline_source": rf["traffic_source"], "pmc_stale": rf["pmc_stale"],
           "stage_ms_per_step": rec["stage_ms_per_step"], "pairs_per_tier": rec["tiers"]["pairs_per_tier"],
           "pairs_banded": rec["tiers"]["pairs_banded"], "auto_budget": rec["tiers"]["auto_budget"],
           "parity_sample": rec["parity_sample"]}
    out.update(h2h_out)
    if not force_band and cpu_leg:
        per_pair_us = 2.0 if wl["length"] <= 200 else 95.0 * (wl["length"] / 1000.0) ** 2
        cb = cpu_baseline(buf, meta, wl["cigar"], int(max(16, min(n, 4e6 / per_pair_us))), pen=pen_of(wl))
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "reference_shim") if k in cb}
    out["leg_s"] = round(time.perf_counter() - t0, 1)
    return out


def summary_of(out):
    """The numbers SURVEY.md section 8(d) asks for, flat and short: resident value, the reference's own metric (wall of
    launch_alignments, PCIe both ways), the CLI's printed wall time, the CPU baselines, every other configuration's value and the main
    kernel's fractions."""
    rf = out.get("roofline") or {}
    cb = out.get("cpu_baseline") or {}
    cw = out.get("cli_wall") or {}
    sm = {"value": out.get("value"), "ms_per_step": out.get("ms_per_step"),
          "host_to_host_value": out.get("host_to_host_value"), "host_to_host_ms_per_call": out.get("host_to_host_ms_per_call"),
          "host_to_host_cold_ms": out.get("host_to_host_cold_ms"), "host_to_host_max_over_median": out.get("host_to_host_max_over_median"),
          "cli_wall_value": out.get("cli_wall_value"), "cli_wall_ms": cw.get("best_wall_ms"), "cli_bring_up_wait_ms": cw.get("bring_up_wait_ms"),
          "cpu_baseline_value": cb.get("value"), "cpu_baseline_cores": cb.get("cores"),
          "cpu_baseline_shim_value": (cb.get("reference_shim") or {}).get("value"),
          "main_kernel_ms": rf.get("kernel_ms"), "cells_per_s": rf.get("cells_per_s"), "valu_busy_frac": rf.get("frac") if rf.get("bound") == "valu-issue" else None,
          "issue_bound_frac": rf.get("issue_bound_frac"), "hbm_algorithmic_frac": rf.get("hbm_algorithmic_frac"),
          "parity_sample_ok": (out.get("parity_sample") or {}).get("bit_exact_vs_oracle"),
          "n_gpus": out.get("n_gpus"), "library_call_value": out.get("library_call_value")}
    for key, leg in (out.get("configs") or {}).items():
        sm[f"{key}_value"] = leg.get("value")
        sm[f"{key}_kernel_ms"] = leg.get("kernel_ms")
        if key.startswith("cfg3_") or key == "cfg2_x5o3e2":
            sm[f"{key}_cells_per_s"] = leg.get("cells_per_s")
            sm[f"{key}_tier"] = leg.get("tier")
        if leg.get("host_to_host_value") is not None:
            sm[f"{key}_host_to_host_value"] = leg.get("host_to_host_value")
        ps = leg.get("parity_sample") or {}
        sm[f"{key}_parity_ok"] = ps.get("bit_exact_vs_oracle", ps.get("valid_and_cost_equals_score"))
    ob = out.get("ont_banded") or {}
    for ds in ("iid_10pct", "long_read_shaped"):
        if isinstance(ob.get(ds), dict):
            sm[f"ont_{ds}_exact_ms"] = ob[ds].get("exact_ms")
    return sm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default: enough steps of the workload for a >= 5 s timed region")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default: the workload's)")
    ap.add_argument("--max-error", type=int, default=0)
    ap.add_argument("--mode", default="ranks", choices=["ranks", "library"])
    ap.add_argument("--h2h-timing", type=int, default=0, help="stage clocks of the host-to-host calls on stderr (1: per call, 2: per batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configurations and the CLI leg")
    ap.add_argument("--ont-banded-only", action="store_true", help="only the ONT-shaped exact-against-banded grid (the reference's published experiment in shape)")
    ap.add_argument("--no-inherit-budgets", action="store_true", help="sample the score budgets again in every step")
    ap.add_argument("--force-band", action="store_true", help="banded workloads: always run the banded kernels (tuning.force_band)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT",
                    help="wfagpu_amd_tuning_t field for A/B runs (waves_per_simd=7, max_blocks_per_cu=24, ...)")
    ap.add_argument("--virtual-devices", type=int, default=0,
                    help="--mode library: shard the call over this many device slots mapped onto the visible GPUs")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1: still run as a torch.distributed job of one rank (self-spawned torchrun child, RCCL process group, "
                         "barrier + max-over-ranks all-reduce + all-gather executed) -- exercises the N > 1 harness on one GPU")
    ap.add_argument("--cpu-harness", action="store_true",
                    help="CPU-only check of the rank/timing harness (gloo; the step is the ORACLE, nothing is measured)")
    args = ap.parse_args()

    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.mode == "ranks" and (args.gpus > 1 or args.force_dist) and not in_torchrun:
        self_spawn(args)

    wl = dict(WORKLOADS[args.workload])
    n_pairs = args.pairs or wl["pairs"]
    max_error = args.max_error or wl["max_error"]
    steps = args.steps or wl["steps"]

    import shardlib
    rank, local_rank, world = shardlib.env_rank_world()
    if args.mode == "ranks" and in_torchrun and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")

    if args.cpu_harness:
        return cpu_harness(args, rank, world, n_pairs, steps)

    import torch
    import wfagpu
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"

    if args.mode == "library":
        return library_mode(args, wl, n_pairs, max_error, steps)
    if args.ont_banded_only:
        print(json.dumps({"ont_banded": ont_banded_leg(full=True)}))
        return

    dist = None
    torch.cuda.set_device(local_rank)
    host_grp = None
    if world > 1 or (args.force_dist and in_torchrun):
        dist = shardlib.init_distributed("nccl", device=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm
        host_grp = shardlib.host_group(dist)

    # synthetic data (seeded; every rank its own shard), resident in HBM before the clock starts
    tuning = {"force_band": 1} if args.force_band else {}
    tuning.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tuning})
    # (the committed counters belong to the default command: not to other sizes, and not to runs with the A/B switches set)
    default_cmd = not args.pairs and not args.max_error and not tuning
    if wl.get("force_band"):
        tuning["force_band"] = 1      # (part of the workload: cfg4b IS configs[3] on the banded kernels)
        args.force_band = True
    tuning.update(wl.get("tuning", {}))      # (switches that are part of a workload: cfg5t3)
    data = make_pairs(WORKLOADS[args.workload].get("pairs_like", args.workload), n_pairs, shardlib.shard_seed(1000, rank), world=world)
    pre = {}
    if rank == 0 and world == 1:
        # The legs that measure COLD calls come first, before this process has allocated (and released) anything big:
        # the CLI as a fresh process, then launch_alignments* from this one (see settle()).
        # (the HIP runtime itself is initialised here, as the CLI's device query does before its clock starts: ~60-90 ms)
        ndev = __import__("ctypes").c_int(0)
        wfagpu.load().get_num_cuda_devices(__import__("ctypes").byref(ndev))
        settle(3.0)
        if default_cmd and args.workload == "cfg3" and not args.no_configs:
            try:
                pre["cli_wall"] = cli_wall()
                pre["cli_wall_value"] = pre["cli_wall"].get("value")
            except Exception as ex:
                pre["cli_wall"] = {"error": str(ex)}
            settle(2.0)
        if not args.no_host_to_host:
            try:
                h2h = host_to_host(data[0], data[1], wl, max_error, tuning=tuning, launch_cfg={"timing": args.h2h_timing} if args.h2h_timing else None)
                pre["host_to_host"] = h2h
                # the reference's own metric (tools/aligner.c:450-474), PCIe inclusive, next to `value` (resident batch)
                pre["host_to_host_value"] = h2h["pageable"]["warm"]
                pre["host_to_host_ms_per_call"] = h2h["pageable"]["warm_ms"]
                pre["host_to_host_cold_ms"] = h2h["pageable"]["cold_ms"]
                pre["host_to_host_max_over_median"] = h2h["pageable"]["max_over_median"]
                if h2h["stages_ms"].get("host_packed_batches"):
                    # the call above packed its sequences on the host (a quarter of the bytes over PCIe); the same call
                    # with the ASCII going up and the pack kernel running, for comparison
                    asc = host_to_host(data[0], data[1], wl, max_error, tuning=tuning, reps=4, launch_cfg={"host_pack": -1})
                    h2h["ascii_upload"] = {"warm_ms": asc["pageable"]["warm_ms"], "best_ms": asc["pageable"]["best_ms"],
                                           "cold_ms": asc["pageable"]["cold_ms"], "warm": asc["pageable"]["warm"],
                                           "stages_ms": asc["stages_ms"]}
            except Exception as ex:
                pre["host_to_host"] = {"error": str(ex)}
    rec, buf, meta = resident_leg(args.workload, n_pairs, max_error, steps, args.warmup, local_rank, shardlib.shard_seed(1000, rank), tuning,
                                  not args.no_inherit_budgets, dist=dist, use_pmc=default_cmd, data=data)
    elapsed = rec["elapsed"]
    band = wl["band"]
    total_pairs = n_pairs * steps * world
    value = total_pairs / elapsed
    gcups = rec["dptt"] * steps * world / elapsed / 1e9

    out = None
    if rank == 0:
        out = {
            "metric": "alignments_per_sec", "value": round(value, 1), "unit": "alignments/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "pairs_per_gpu_per_step": n_pairs, "length": wl["length"],
                       "error": wl["error"], "penalties": "x=%d,o=%d,e=%d" % pen_of(wl), "max_error": max_error,
                       "compute_cigar": wl["cigar"],
                       "budgets": "tuned on a sample in the warm-up, inherited by the timed steps (same stream of reads)" if rec["inherit"]
                                  else "tuned on a sample in every step",
                       "band": {"period": band[0], "width": band[1], "forced": bool(args.force_band),
                                "policy": "the band is used only where the sampled score budgets leave the exact wavefronts wider "
                                          "than 1.75 bands (tiers.pairs_banded counts the pairs it finished)"} if band else None,
                       "sharding": f"batch-sharded x{world}, no collective", "mode": "ranks",
                       "process_group": ("rccl (backend nccl), world size %d: barrier + max all-reduce + all-gather of the clocks" % world) if dist is not None else None},
            "gcups": round(gcups, 2),
            # every rank's own clock over the same K steps (the aggregate above uses the slowest: max over ranks)
            "per_rank": [{"rank": r, "ms_per_step": round(e / steps * 1e3, 3), "value": round(n_pairs * steps / e, 1)}
                         for r, e in enumerate(rec["per_rank_elapsed"])],
            "stage_ms_per_step": rec["stage_ms_per_step"],
            "tiers": rec["tiers"],
            "roofline": rec["roofline"],
            "parity_sample": rec["parity_sample"],
        }
    if rank == 0 and world == 1:
        out.update(pre)
        if not args.no_cpu_baseline:
            # ~10-30 s of CPU work: bounded sample of the same workload
            per_pair_us = 2.0 if wl["length"] <= 200 else 95.0 * (wl["length"] / 1000.0) ** 2
            budget = int(max(16, min(n_pairs, 20e6 / per_pair_us)))
            out["cpu_baseline"] = cpu_baseline(buf, meta, wl["cigar"], budget, pen=pen_of(wl))
        if default_cmd and args.workload == "cfg3" and not args.no_configs:
            out["configs"] = {}
            # configs[2] under the reference's other penalty sets, on the headline's own pairs (resident legs only)
            for key, name, k, w in (("cfg3_x5o3e2", "cfg3_x5o3e2", 6, 2), ("cfg3_x3o1e4", "cfg3_x3o1e4", 6, 2)):
                try:
                    out["configs"][key] = extra_config(name, k, w, data=(buf, meta), h2h_leg=False, cpu_leg=False)
                except Exception as ex:
                    out["configs"][key] = {"error": str(ex)}
            del buf, meta, data
            # the other BASELINE GPU configurations, short legs (~60 s together)
            for key, name, k, w, h2h_leg, cpu_leg in (("cfg2", "cfg2", 400, 5, True, True), ("cfg2_with_cigar", "cfg2c", 200, 5, True, True),
                                                      ("cfg2_x5o3e2", "cfg2_x5o3e2", 400, 5, False, False),
                                                      ("cfg4", "cfg4", 8, 3, True, True), ("cfg4_band_forced", "cfg4b", 8, 3, True, True),
                                                      ("cfg5", "cfg5", 4, 2, True, True)):
                try:
                    out["configs"][key] = extra_config(name, k, w, h2h_leg=h2h_leg, cpu_leg=cpu_leg)
                except Exception as ex:
                    out["configs"][key] = {"error": str(ex)}
            for key, leg in out["configs"].items():      # (flat copies: the driver's record keeps top-level scalars)
                out[f"{key}_value"] = leg.get("value")
            try:
                out["ont_banded"] = ont_banded_leg()
                rows = out["ont_banded"]["long_read_shaped"]["rows"]
                r1024 = [r for r in rows if r["beta"] == 1024 and r["lambda"] == 100]
                if r1024:      # (flat: the reference's suggested operating point for a large band, lambda = 100)
                    out["ont_banded_beta1024_lambda100_speedup"] = r1024[0]["speedup"]
                    out["ont_banded_beta1024_lambda100_recall"] = r1024[0]["recall"]
            except Exception as ex:
                out["ont_banded"] = {"error": str(ex)}
        # (the LAST key of the line -- the driver's record keeps the tail of a long line -- is `summary`, the contract's numbers in one
        # flat object: added where the line is printed)
    if dist is not None:
        # N > 1: the ranks above never share anything but the barrier.  The reference's user calls launch_alignments()
        # ONCE and the library shards the call over the N devices from one process: host RAM bandwidth, the PCIe root and
        # the scatter threads are shared then.  Rank 0 measures that too, on N x P pairs, while the other
        # ranks (their contexts closed above) wait at the barrier; it never replaces `value`.
        # (the wait is a gloo barrier: an RCCL barrier would keep a spinning kernel on every waiting rank's GPU)
        # (... and the call runs in a CHILD process of rank 0 -- `bench.py --mode library --gpus N` -- so that nothing it
        # does, not even a fatal device error, can cost the ranks' own line)
        def library_leg():
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                    "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", "library", "--gpus", str(world), "--workload", args.workload,
                   "--pairs", str(n_pairs), "--max-error", str(max_error), "--steps", "4", "--warmup", "2"]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"error": f"exit code {r.returncode}: {r.stderr[-400:]}"}
            return json.loads(lines[-1])
        if not args.no_host_to_host:
            try:
                leg = shardlib.rank0_exclusive(dist, host_grp, rank, library_leg)
                if rank == 0:
                    out["library_call"] = leg
                    out["library_call_value"] = leg.get("value")
            except Exception as ex:
                if rank == 0:
                    out["library_call"] = {"error": str(ex)}
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out.pop("summary", None)
        out["summary"] = summary_of(out)      # (the last key, whatever was added since)
        print(json.dumps(out))


def device_topology(n):
    """PCIe bus id and NUMA node of the first n HIP devices (sysfs): with the per-device stage times, what makes the first record of
    a real multi-GPU node diagnosable from its line alone (which devices share a root, which host threads sat on which node)."""
    import ctypes as C
    import wfagpu
    out = []
    try:
        hip = wfagpu._hiprt()
        hip.hipDeviceGetPCIBusId.argtypes = [C.c_char_p, C.c_int, C.c_int]
        hip.hipDeviceGetPCIBusId.restype = C.c_int
    except Exception as ex:
        return [{"error": str(ex)}]
    for d in range(n):
        b = C.create_string_buffer(64)
        rc = hip.hipDeviceGetPCIBusId(b, 64, d)
        bdf = b.value.decode().lower() if rc == 0 else None
        node = None
        if bdf:
            try:
                node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
            except (OSError, ValueError):
                node = None
        out.append({"device": d, "pcie_bdf": bdf, "numa_node": node})
    return out


def library_mode(args, wl, n_pairs, max_error, steps):
    """ONE process, N devices inside the library: a step = one launch_alignments() call on N*P pairs in a pageable host
    buffer (H2D, kernels, D2H and the scatter into the caller's result records all inside the clock)."""
    import ctypes as C
    import wfagpu
    n = n_pairs * args.gpus
    buf, meta = wfagpu.generate_pairs(n, wl["length"], wl["error"], seed=1000, nthreads=min(16, usable_cores()))
    lib = wfagpu.load()
    res = C.POINTER(wfagpu.AlignmentResult)()
    assert lib.initialize_wfa_results(C.byref(res), n, 256 if wl["cigar"] else 1)
    band = wl["band"]
    opt = wfagpu.Options(max_error=max_error, threads_per_block=band[1] if band else 64, num_workers=0,
                         band=band[0] if band else -1, batch_size=n, num_alignments=n,
                         penalties=wfagpu.Penalties(*pen_of(wl)), compute_cigar=wl["cigar"])
    fn = lib.launch_alignments if wl["cigar"] else lib.launch_alignments_distance
    wfagpu.configure_launch(num_devices=args.gpus, virtual_devices=args.virtual_devices,
                            tuning={"force_band": 1} if args.force_band else {})
    nd = C.c_int(0)
    lib.get_num_cuda_devices(C.byref(nd))
    if nd.value < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {nd.value} devices are visible")
    if args.gpus > 1:
        lib.wfagpu_amd_warmup()      # (the device query above brings up the current device only: every device of the call, in the background)
    steps = args.steps or max(5, steps // 5)
    for _ in range(max(1, args.warmup)):
        fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    elapsed = time.perf_counter() - t0
    stages = wfagpu.last_launch_stats()
    per_dev = [{"slot": i, "device": d["devices"], "wall_ms": round(d["total_ms"], 2), "host_threads": d["host_threads"], "lanes": d["lanes"],
                "batches": d["batches"], "sequences": "packed on the host" if d["host_packed_batches"] == d["batches"] else
                ("ASCII, packed by the kernel" if d["host_packed_batches"] == 0 else f"{d['host_packed_batches']} of {d['batches']} batches packed on the host"),
                "host_pack_threads": d["host_pack_threads"], "upload_ms": round(d["upload_ms"], 2), "device_ms": round(d["device_ms"], 2),
                "device_wait_ms": round(d["device_wait_ms"], 2), "scatter_ms": round(d["scatter_ms"], 2)}
               for i, d in enumerate(wfagpu.last_launch_stats_per_device())]
    topo = {t.get("device"): t for t in device_topology(nd.value)}
    for d in per_dev:      # (where the slot's device sits: PCIe bus id, NUMA node)
        d.update({k: v for k, v in (topo.get(d["device"]) or {}).items() if k != "device"})
    dptt = int((meta["pattern_len"].astype(np.int64) * meta["text_len"].astype(np.int64)).sum())
    out = {"metric": "alignments_per_sec", "value": round(n * steps / elapsed, 1), "unit": "alignments/s",
           "host_cores": usable_cores(), "host_numa_nodes": len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]) if os.path.isdir("/sys/devices/system/node") else None,
           "n_gpus": args.gpus, "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {wl['desc']}", "pairs_per_gpu_per_step": n_pairs, "length": wl["length"],
                      "error": wl["error"], "penalties": "x=%d,o=%d,e=%d" % pen_of(wl), "max_error": max_error, "compute_cigar": wl["cigar"],
                      "sharding": f"one launch_alignments() call sharded in-library over {args.gpus} devices, no collective",
                      "mode": "library (host buffer -> host results, PCIe inclusive)"},
           "gcups": round(dptt * steps / elapsed / 1e9, 2), "first_result": int(res[0].error),
           "virtual_devices": args.virtual_devices, "per_device_last_call": per_dev,
           "stages_ms_last_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stages.items()}}
    lib.destroy_wfa_results(res, n)
    print(json.dumps(out))


def cpu_harness(args, rank, world, n_pairs, steps):
    """No GPU: the same rank discovery, barrier + max-over-ranks timing and rank-0 aggregation with gloo, the ORACLE as
    the step.  Exists so that the N>1 launch path (including the self-spawn) is covered by CPU tests; it measures nothing."""
    import oracle_lib
    import shardlib
    import wfagpu
    dist = shardlib.init_distributed("gloo") if (world > 1 or (args.force_dist and "WORLD_SIZE" in os.environ)) else None
    n = min(n_pairs, 64)
    gen_threads = max(1, min(16, usable_cores() // max(1, world)))      # (what make_pairs gives a rank of the GPU run)
    buf, meta = wfagpu.generate_pairs(n, 120, 0.05, seed=shardlib.shard_seed(1000, rank), nthreads=gen_threads)
    state = {}

    def step():
        state["s"], _, _ = oracle_lib.oracle_batch(buf, meta, PEN, cigar=False)

    elapsed, per_rank = shardlib.timed_steps(step, steps, args.warmup, dist=dist, per_rank=True)
    excl = None
    if dist is not None:
        # the rank-0-alone leg of the GPU run (the in-library sharded call), with a stand-in for the call
        grp = shardlib.host_group(dist)
        excl = shardlib.rank0_exclusive(dist, grp, rank, lambda: {"ranks_parked": world - 1})
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "alignments_per_sec", "harness_only": True, "n_gpus": world, "steps": steps,
                          "warmup": args.warmup, "value": n * steps * world / elapsed, "unit": "alignments/s",
                          "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "library_call": excl, "generator_threads_per_rank": gen_threads,
                          "per_rank": [{"rank": r, "ms_per_step": e / steps * 1e3, "value": n * steps / e} for r, e in enumerate(per_rank)]}))


if __name__ == "__main__":
    main()
