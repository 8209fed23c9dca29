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
  roofline      the dominant kernel's main launch against the HBM roofline (SURVEY.md 8d algorithmic bytes) and, from the
                committed PMC pass of the same command, against the instruction-issue limit that really binds it
  cpu_baseline  the reference's WFA2 (oracle/_ref) on this box's host cores, bounded sample, N=1 only
  host_to_host  the reference's own metric: wall of launch_alignments from a pageable host buffer to host CIGAR buffers
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
    "cfg3": dict(pairs=1_000_000, length=1000, error=0.05, cigar=True, max_error=300, band=None, steps=100,
                 desc="1M synthetic 1 kbp pairs, 5% error, x=2,o=3,e=1, score+CIGAR"),
    "cfg4": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=(25, 512), steps=200,
                 desc="16k HiFi-shaped 10 kbp pairs, 3% error, -B auto (re-centre every 25 scores) -t 512 banded, score+CIGAR"),
    "cfg4x": dict(pairs=16_384, length=10_000, error=0.03, cigar=True, max_error=3000, band=None, steps=100,
                  desc="16k HiFi-shaped 10 kbp pairs, 3% error, exact (unbanded), score+CIGAR"),
    "cfg5": dict(pairs=1024, length=30_000, error=0.10, cigar=True, max_error=9000, band=None, steps=25,
                 desc="1024 ONT-shaped 30 kbp pairs, 10% error, exact (unbanded), -e 9000, score+CIGAR"),
}
PEN = (2, 3, 1)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def kernel_source_hash():
    """Identifies the kernels a PMC summary was taken with (profiles/*/..._pmc_counters.csv carry it in a '#' line)."""
    h = hashlib.sha1()
    for f in ("align_kernel.hip", "trace_kernel.hip", "pack_kernel.hip", "wfa_device.h"):
        h.update(open(os.path.join(ROOT, "wfa-gpu_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(seq_bytes, pairs, cells, compute_cigar):
    """SURVEY.md section 8(d), share of the wavefront kernel: packed sequences read + 48 B record + 20 B result per pair,
    + 6 B per wavefront cell (three 16-bit offsets) in CIGAR mode."""
    return int(seq_bytes + 68 * pairs + (6 * cells if compute_cigar else 0))


def _pmc_main_launch(workload, counters, optional=()):
    """Counters of the MAIN launch (largest WRITE_SIZE+FETCH_SIZE) of wfa_align_kernel from the newest committed rocprofv3
    PMC summary of this same command (profiles/rNN/<workload>_pmc_counters.csv; separate --pmc passes over one step).
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
    rows = [r for r in csv.DictReader(ln for ln in lines if not ln.startswith("#")) if "wfa_align_kernel" in r["kernel"]]
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


def cpu_baseline(buf, meta, compute_cigar, budget_pairs):
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
        run = lambda m, nt: oracle_lib.ref_batch(sub, m, PEN, cigar=compute_cigar, memory_mode=1, nthreads=nt)
    else:
        kind = "port"
        run = lambda m, nt: oracle_lib.oracle_batch(sub, m, PEN, cigar=compute_cigar, nthreads=nt)
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
                         penalties=wfagpu.Penalties(*PEN), compute_cigar=wl["cigar"])
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
    calls = timed(3 + reps)
    ms = [c[0] for c in calls]
    steady = sorted(calls[3:], key=lambda c: c[0])
    med_ms, med_stats = steady[len(steady) // 2]
    out = {"unit": "alignments/s", "pairs": n, "devices": n_devices,
           "what": "wall of launch_alignments%s(): pageable host buffer -> device -> host results%s" %
                   ("" if wl["cigar"] else "_distance", " with CIGAR strings" if wl["cigar"] else ""),
           "pageable": {"cold_ms": round(ms[0], 2), "warm_ms": round(med_ms, 2), "best_ms": round(steady[0][0], 2),
                        "cold": round(n / ms[0] * 1e3, 1), "warm": round(n / med_ms * 1e3, 1),
                        "best": round(n / steady[0][0] * 1e3, 1), "calls_ms": [round(m, 2) for m in ms]},
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
    ap.add_argument("--no-inherit-budgets", action="store_true", help="sample the score budgets again in every step")
    ap.add_argument("--force-band", action="store_true", help="banded workloads: always run the banded kernels (tuning.force_band)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT",
                    help="wfagpu_amd_tuning_t field for A/B runs (waves_per_simd=7, max_blocks_per_cu=24, ...)")
    ap.add_argument("--virtual-devices", type=int, default=0,
                    help="--mode library: shard the call over this many device slots mapped onto the visible GPUs")
    ap.add_argument("--cpu-harness", action="store_true",
                    help="CPU-only check of the rank/timing harness (gloo; the step is the ORACLE, nothing is measured)")
    args = ap.parse_args()

    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.mode == "ranks" and args.gpus > 1 and not in_torchrun:
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

    dist = None
    torch.cuda.set_device(local_rank)
    host_grp = None
    if world > 1:
        dist = shardlib.init_distributed("nccl", device=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm
        host_grp = shardlib.host_group(dist)

    # synthetic data (seeded; every rank its own shard), resident in HBM before the clock starts
    buf, meta = wfagpu.generate_pairs(n_pairs, wl["length"], wl["error"], seed=shardlib.shard_seed(1000, rank),
                                      nthreads=min(16, usable_cores()))
    tuning = {"force_band": 1} if args.force_band else {}
    tuning.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tuning})
    al = wfagpu.DeviceAligner(local_rank, **tuning)
    batch = al.upload(buf, meta)
    dptt = int((meta["pattern_len"].astype(np.int64) * meta["text_len"].astype(np.int64)).sum())
    acc = {"align_ms": 0.0, "pack_ms": 0.0, "trace_ms": 0.0, "launches": 0, "main_ms": 0.0}
    last = {}
    band = wl["band"]

    def step():
        last["out"] = al.align(batch, PEN, max_error=max_error, compute_cigar=wl["cigar"], band=band[0] if band else -1,
                               band_width=band[1] if band else 0, fetch=False)
        st = al.stats()
        acc["align_ms"] += st.align_ms
        acc["pack_ms"] += st.pack_ms
        acc["trace_ms"] += st.trace_ms
        acc["launches"] += st.align_launches
        acc["main_ms"] += st.main_launch_ms

    for _ in range(args.warmup):
        step()
    # The steps of a run are batches of one stream of reads: like launch_alignments* does for the batches of a call, the
    # score budgets tuned on a sample during the warm-up are tried again without sampling (results stay exact: a pair that
    # misses its budget is re-run, and the parity sample below is taken from the last timed step).
    inherit = args.warmup > 0 and not args.no_inherit_budgets
    if inherit:
        al.hint_same_stream(True)
    for k in acc:
        acc[k] = 0
    elapsed = shardlib.timed_steps(step, steps, 0, dist=dist, sync=torch.cuda.synchronize, device="cuda")
    d_scores, ptrs = last["out"]

    st = al.stats()
    total_pairs = n_pairs * steps * world
    value = total_pairs / elapsed
    gcups = dptt * steps * world / elapsed / 1e9

    out = None
    if rank == 0:
        # Dominant kernel = wfa_align_kernel; its MAIN launch (a step also has the short launches of the auto-budget
        # sample and of re-runs).  Duration: HIP events on the stream the kernel is launched on (csrc/wfa_host.hip),
        # averaged over the timed steps.
        main_ms = acc["main_ms"] / steps
        alg = algorithmic_bytes(int(st.main_launch_seq_bytes), int(st.main_launch_pairs), int(st.main_launch_cells), wl["cigar"])
        achieved = alg / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
        # (the committed counters belong to the default command: not to other sizes, and not to runs with the A/B switches set)
        default_cmd = not args.pairs and not args.max_error and not tuning
        hw = ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
              "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"]
        pmc, pmc_src, pmc_stale = _pmc_main_launch(args.workload, ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"] + hw, optional=hw) \
            if default_cmd else (None, None, None)
        # HBM bytes of the main launch: (FETCH_SIZE*2 + WRITE_SIZE) KB -- FETCH_SIZE counts half of a wide coalesced
        # read on gfx950 (MI355X_MICROARCH.md, HBM section)
        traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024) if pmc else None
        roofline = {"bound": "valu-issue", "kernel": "wfa_align_kernel (main launch of a step)",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": pmc_src,
                    "pmc_stale": pmc_stale,
                    "algorithmic_bytes_per_launch": alg, "kernel_ms": round(main_ms, 4),
                    "launch": {"tier": int(st.main_launch_tier), "pairs": int(st.main_launch_pairs),
                               "cells": int(st.main_launch_cells)},
                    "launches_per_step": acc["launches"] / steps,
                    "all_launches_ms_per_step": round(acc["align_ms"] / steps, 4),
                    "cells_per_step": int(st.cells),
                    "cells_per_s": round(st.main_launch_cells / (main_ms * 1e-3), 1) if main_ms > 0 else None,
                    "note": "achieved/peak/frac: SURVEY 8(d) algorithmic bytes of the main launch / its HIP-event duration "
                            "vs HBM peak.  The kernel keeps its wavefronts in LDS and stores 1 B per cell, so HBM is not "
                            "what binds it: see `issue` (instruction issue, from the committed PMC pass) and DESIGN.md"}
        if pmc and main_ms > 0:
            # What this kernel really saturates: instruction issue.  A SIMD issues at most one vector and one scalar
            # instruction per 4 cycles (profiles/r02/valu_rate.txt: 4.0-4.2 cycles per integer wave64 op); 1024 SIMDs, 2.4 GHz.
            peak = 1024 * 0.25 * 2.4e9 / 1e9
            roofline["issue"] = {"unit": "G wave-instr/s", "peak_per_pipe": round(peak, 1), "source": pmc_src, "stale": pmc_stale}
            for pipe, key in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU")):
                ach = pmc[key] / (main_ms * 1e-3) / 1e9
                roofline["issue"][pipe] = {"achieved": round(ach, 1), "frac": round(ach / peak, 3), "insts": int(pmc[key])}
            if "SQ_ACTIVE_INST_VALU" in pmc and pmc.get("GRBM_GUI_ACTIVE"):
                # The same from the hardware's own busy counters (the gfx9 VALUBusy / SALUBusy formulas): SQ_ACTIVE_INST_*
                # count quad-cycles a pipe spent on instructions, summed over the chip; GRBM_GUI_ACTIVE is summed over the 8
                # XCDs.  No microbenchmark in the denominator.
                simds, xcds = 1024, 8
                cyc = pmc["GRBM_GUI_ACTIVE"] / xcds
                roofline["issue"]["hw_counters"] = {
                    "valu_busy": round(pmc["SQ_ACTIVE_INST_VALU"] * 4 / simds / cyc, 3),
                    "salu_busy": round(pmc["SQ_ACTIVE_INST_SCA"] * 4 / simds / cyc, 3),
                    "lds_busy": round(pmc.get("SQ_ACTIVE_INST_LDS", 0) * 4 / simds / cyc, 3),
                    "wave_cycles_split": {k: round(pmc[c] / pmc["SQ_WAVE_CYCLES"], 3) for k, c in
                                          (("issuing", "SQ_ACTIVE_INST_ANY"), ("stalled_at_issue", "SQ_WAIT_INST_ANY"), ("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"))
                                          if pmc.get("SQ_WAVE_CYCLES") and c in pmc},
                    "clock_ghz_during_launch": round(cyc / (pmc["_ms_GRBM_GUI_ACTIVE"] * 1e-3) / 1e9, 3) if pmc.get("_ms_GRBM_GUI_ACTIVE") else None,
                    "formula": "SQ_ACTIVE_INST_{VALU,SCA,LDS} * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
        out = {
            "metric": "alignments_per_sec", "value": round(value, 1), "unit": "alignments/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "pairs_per_gpu_per_step": n_pairs, "length": wl["length"],
                       "error": wl["error"], "penalties": "x=2,o=3,e=1", "max_error": max_error,
                       "compute_cigar": wl["cigar"],
                       "budgets": "tuned on a sample in the warm-up, inherited by the timed steps (same stream of reads)" if inherit
                                  else "tuned on a sample in every step",
                       "band": {"period": band[0], "width": band[1], "forced": bool(args.force_band),
                                "policy": "the band is used only where the sampled score budgets leave the exact wavefronts wider "
                                          "than 2.5 bands (tiers.pairs_banded counts the pairs it finished)"} if band else None,
                       "sharding": f"batch-sharded x{world}, no collective", "mode": "ranks"},
            "gcups": round(gcups, 2),
            "stage_ms_per_step": {"pack": round(acc["pack_ms"] / steps, 3), "align": round(acc["align_ms"] / steps, 3),
                                  "trace": round(acc["trace_ms"] / steps, 3)},
            "tiers": {"lds_bytes_first": int(st.lds_bytes_tier0), "blocks_per_cu_first": int(st.blocks_per_cu_tier0), "waves_per_simd_first": int(st.waves_per_simd_tier0),
                      "pairs_per_tier": [int(v) for v in st.pairs_tier], "pairs_retried": int(st.pairs_retried),
                      "pairs_banded": int(st.pairs_banded), "auto_budget": int(st.auto_budget),
                      "pairs_budget_missed": int(st.pairs_budget_missed), "passes": int(st.sub_batches),
                      "arena_gb": round(st.arena_units * 16 / 1e9, 2)},
            "roofline": roofline,
        }
        # parity spot check outside the timed region: a sample against the checker
        try:
            import oracle_lib
            k = min(2000 if wl["length"] <= 1000 else (64 if wl["length"] <= 10000 else 8), n_pairs)
            scores = d_scores[:k].cpu().numpy()
            if oracle_lib.have_ref():
                so, co = oracle_lib.ref_batch(buf, meta[:k], PEN, cigar=wl["cigar"], memory_mode=0, nthreads=min(16, usable_cores()))
            else:
                so, co, _ = oracle_lib.oracle_batch(buf, meta[:k], PEN, cigar=wl["cigar"], nthreads=min(8, usable_cores()))
            cg = wfagpu.fetch_cigars(ptrs[0], ptrs[1], ptrs[2], n_pairs, st.text_bytes)[:k] if wl["cigar"] else None
            if band:
                # the adaptive band is a heuristic: valid alignments, cost == score >= optimum; recall reported
                pairs = wfagpu.pairs_from_layout(buf, meta[:k])
                chk = [oracle_lib.check_cigar(p, t, c, PEN) for (p, t), c in zip(pairs, cg)]
                ok = all(o and cost == s for (o, cost), s in zip(chk, scores)) and bool((scores >= so).all())
                out["parity_sample"] = {"pairs": k, "valid_and_cost_equals_score": ok, "recall": float((scores == so).mean())}
            else:
                ok = bool(np.array_equal(scores, so)) and (not wl["cigar"] or cg == co)
                out["parity_sample"] = {"pairs": k, "bit_exact_vs_oracle": ok}
        except Exception as ex:  # the checker is optional for the measurement itself
            out["parity_sample"] = {"error": str(ex)}
    al.close()
    del batch
    torch.cuda.empty_cache()
    torch.cuda.synchronize()      # (the resident leg's device memory is back with the driver before the host-to-host leg starts)
    if rank == 0 and world == 1:
        if not args.no_host_to_host:
            try:
                h2h = host_to_host(buf, meta, wl, max_error, tuning=tuning, launch_cfg={"timing": args.h2h_timing} if args.h2h_timing else None)
                out["host_to_host"] = h2h
                # the reference's own metric (tools/aligner.c:450-474), PCIe inclusive, next to `value` (resident batch)
                out["host_to_host_value"] = h2h["pageable"]["warm"]
                out["host_to_host_ms_per_call"] = h2h["pageable"]["warm_ms"]
                if h2h["stages_ms"].get("host_packed_batches"):
                    # the call above packed its sequences on the host (a quarter of the bytes over PCIe); the same call
                    # with the ASCII going up and the pack kernel running, for comparison
                    asc = host_to_host(buf, meta, wl, max_error, tuning=tuning, reps=4, launch_cfg={"host_pack": -1})
                    h2h["ascii_upload"] = {"warm_ms": asc["pageable"]["warm_ms"], "best_ms": asc["pageable"]["best_ms"],
                                           "cold_ms": asc["pageable"]["cold_ms"], "warm": asc["pageable"]["warm"],
                                           "stages_ms": asc["stages_ms"]}
            except Exception as ex:
                out["host_to_host"] = {"error": str(ex)}
        if not args.no_cpu_baseline:
            # ~10-30 s of CPU work: bounded sample of the same workload
            per_pair_us = 2.0 if wl["length"] <= 200 else 95.0 * (wl["length"] / 1000.0) ** 2
            budget = int(max(16, min(n_pairs, 20e6 / per_pair_us)))
            out["cpu_baseline"] = cpu_baseline(buf, meta, wl["cigar"], budget)
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
        print(json.dumps(out))


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
                         penalties=wfagpu.Penalties(*PEN), compute_cigar=wl["cigar"])
    fn = lib.launch_alignments if wl["cigar"] else lib.launch_alignments_distance
    wfagpu.configure_launch(num_devices=args.gpus, virtual_devices=args.virtual_devices,
                            tuning={"force_band": 1} if args.force_band else {})
    nd = C.c_int(0)
    lib.get_num_cuda_devices(C.byref(nd))
    if nd.value < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but only {nd.value} devices are visible")
    steps = args.steps or max(5, steps // 5)
    for _ in range(max(1, args.warmup)):
        fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(buf.ctypes.data, buf.nbytes, meta.ctypes.data, res, opt, False)
    elapsed = time.perf_counter() - t0
    stages = wfagpu.last_launch_stats()
    dptt = int((meta["pattern_len"].astype(np.int64) * meta["text_len"].astype(np.int64)).sum())
    out = {"metric": "alignments_per_sec", "value": round(n * steps / elapsed, 1), "unit": "alignments/s",
           "n_gpus": args.gpus, "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {wl['desc']}", "pairs_per_gpu_per_step": n_pairs, "length": wl["length"],
                      "error": wl["error"], "penalties": "x=2,o=3,e=1", "max_error": max_error, "compute_cigar": wl["cigar"],
                      "sharding": f"one launch_alignments() call sharded in-library over {args.gpus} devices, no collective",
                      "mode": "library (host buffer -> host results, PCIe inclusive)"},
           "gcups": round(dptt * steps / elapsed / 1e9, 2), "first_result": int(res[0].error),
           "virtual_devices": args.virtual_devices,
           "stages_ms_last_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stages.items()}}
    lib.destroy_wfa_results(res, n)
    print(json.dumps(out))


def cpu_harness(args, rank, world, n_pairs, steps):
    """No GPU: the same rank discovery, barrier + max-over-ranks timing and rank-0 aggregation with gloo, the ORACLE as
    the step.  Exists so that the N>1 launch path (including the self-spawn) is covered by CPU tests; it measures nothing."""
    import oracle_lib
    import shardlib
    import wfagpu
    dist = shardlib.init_distributed("gloo") if world > 1 else None
    n = min(n_pairs, 64)
    buf, meta = wfagpu.generate_pairs(n, 120, 0.05, seed=shardlib.shard_seed(1000, rank))
    state = {}

    def step():
        state["s"], _, _ = oracle_lib.oracle_batch(buf, meta, PEN, cigar=False)

    elapsed = shardlib.timed_steps(step, steps, args.warmup, dist=dist)
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
                          "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "library_call": excl}))


if __name__ == "__main__":
    main()
