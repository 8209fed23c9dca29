#!/usr/bin/env python3
"""Headline benchmark of the MI355X gap-affine WFA path (BASELINE.json metric: alignments/s + GCUPS).

One "step" = one pass of the whole hot path (2-bit pack -> wavefront kernels -> backtrace -> CIGAR text)
over one batch that is already resident in HBM.  Default workload = BASELINE.json configs[2], the
configuration the north-star target is quoted on: 1M synthetic 1 kbp pairs at 5 % error, penalties
(2,3,1), score + CIGAR.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3] [--pairs P] [--max-error E]

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the batch is sharded -- every rank
aligns its own P pairs (weak scaling), no data-path collective; RCCL only carries the barrier and the
max-over-ranks of the timing.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (pairs, length, error, compute_cigar, default max_error, description)
    "cfg2": (100_000, 150, 0.02, False, 45, "100k synthetic 150 bp pairs, 2% error, x=2,o=3,e=1, score-only"),
    "cfg3": (1_000_000, 1000, 0.05, True, 300, "1M synthetic 1 kbp pairs, 5% error, x=2,o=3,e=1, score+CIGAR"),
}
PEN = (2, 3, 1)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(meta, cells, cigar_bytes, compute_cigar):
    """in + out + bt per pair, summed over the batch.
       in  = pad4(P+1)+pad4(T+1) ASCII read + 2*4*(ceil(P/16)+ceil(T/16)) packed write+read + 48 B record
       out = 20 B result (+ CIGAR text), bt = 6 B per wavefront cell in CIGAR mode."""
    P = meta["pattern_len"].astype(np.int64)
    T = meta["text_len"].astype(np.int64)
    pad4 = lambda v: v + (4 - v % 4)
    ascii_b = int((pad4(P + 1) + pad4(T + 1)).sum())
    packed_b = int((4 * ((P + 15) // 16 + (T + 15) // 16)).sum())
    n = len(meta)
    total = ascii_b + 2 * packed_b + 48 * n + 20 * n
    # share of the dominant (wavefront) kernel: packed read + record + result (+ backtrace stream)
    kernel = packed_b + 48 * n + 20 * n
    if compute_cigar:
        total += 6 * cells + cigar_bytes
        kernel += 6 * cells
    return total, kernel


def _pmc_per_launch(workload, counters):
    """Per-launch averages of `counters` for wfa_align_kernel from the newest committed rocprofv3 PMC summary of this
    same command (profiles/rNN/<workload>_pmc_counters.csv; separate --pmc passes over one step).  A step launches
    the kernel several times (sample, main, retries): values are summed over the launches of the pass and divided
    by their number, the same averaging as `kernel_ms_avg`."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_pmc_counters.csv")))
    if not files:
        return None, None
    tot = {c: 0.0 for c in counters}
    cnt = {c: 0 for c in counters}
    for r in csv.DictReader(open(files[-1])):
        if "wfa_align_kernel" in r["kernel"] and r["counter"] in tot:
            tot[r["counter"]] += float(r["value"])
            cnt[r["counter"]] += 1
    if any(cnt[c] == 0 for c in counters):
        return None, None
    return {c: tot[c] / cnt[c] for c in counters}, os.path.relpath(files[-1], ROOT)


def pmc_traffic_bytes(workload):
    """HBM bytes per launch of the dominant kernel: (FETCH_SIZE*2 + WRITE_SIZE) KB -- FETCH_SIZE counts half of a
    wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section).  None if no summary is committed."""
    v, src = _pmc_per_launch(workload, ["FETCH_SIZE", "WRITE_SIZE"])
    if v is None:
        return None, None
    return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), src


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
    import wfagpu
    n = min(len(meta), budget_pairs)
    pairs_meta = meta[:n]
    end = int(max(pairs_meta["text_offset"].max() + pairs_meta["text_len"].max(),
                  pairs_meta["pattern_offset"].max() + pairs_meta["pattern_len"].max())) + 8
    sub = np.ascontiguousarray(buf[:end])
    cores = usable_cores()
    if oracle_lib.have_ref():
        kind = "reference"
        run = lambda: oracle_lib.ref_batch(sub, pairs_meta, PEN, cigar=compute_cigar, memory_mode=1, nthreads=cores)
    else:
        kind = "port"
        run = lambda: oracle_lib.oracle_batch(sub, pairs_meta, PEN, cigar=compute_cigar, nthreads=cores)
    reps = 0
    t0 = time.perf_counter()
    while True:
        out = run()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 2.0 or reps >= 200:
            break
    res = {"value": n * reps / dt, "unit": "alignments/s", "cores": cores, "kind": kind,
           "sample": f"{n} pairs of the same workload x{reps}, {'score+CIGAR' if compute_cigar else 'score-only'}, "
                     f"one aligner per thread, {dt:.2f} s wall"}
    # the same code on ONE core (SURVEY.md section 8d asks for both), on a sample sized for ~1-2 s
    n1 = max(1, min(n, int(n * 1.5 / max(dt / reps, 1e-6) / max(cores, 1))))
    m1 = meta[:n1]
    if kind == "reference":
        run1 = lambda: oracle_lib.ref_batch(sub, m1, PEN, cigar=compute_cigar, memory_mode=1, nthreads=1)
    else:
        run1 = lambda: oracle_lib.oracle_batch(sub, m1, PEN, cigar=compute_cigar, nthreads=1)
    t1 = time.perf_counter()
    run1()
    d1 = time.perf_counter() - t1
    res["single_core"] = {"value": n1 / d1, "unit": "alignments/s", "sample": f"{n1} pairs, {d1:.2f} s wall"}
    return res, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default: the workload's)")
    ap.add_argument("--max-error", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import wfagpu

    import shardlib
    rank, local_rank, world = shardlib.env_rank_world()
    dist = None
    torch.cuda.set_device(local_rank) if torch.cuda.is_available() else None
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if world > 1:
        dist = shardlib.init_distributed("nccl", device=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    n_pairs, length, err, cigar, max_error, desc = WORKLOADS[args.workload]
    if args.pairs:
        n_pairs = args.pairs
    if args.max_error:
        max_error = args.max_error

    # synthetic data (seeded; every rank its own shard), resident in HBM before the clock starts
    buf, meta = wfagpu.generate_pairs(n_pairs, length, err, seed=shardlib.shard_seed(1000, rank),
                                      nthreads=min(16, usable_cores()))
    al = wfagpu.DeviceAligner(local_rank)
    batch = al.upload(buf, meta)
    dptt = int((meta["pattern_len"].astype(np.int64) * meta["text_len"].astype(np.int64)).sum())
    acc = {"align_ms": 0.0, "pack_ms": 0.0, "trace_ms": 0.0, "launches": 0, "steps": 0}
    last = {}

    def step():
        last["out"] = al.align(batch, PEN, max_error=max_error, compute_cigar=cigar, fetch=False)
        st = al.stats()
        acc["align_ms"] += st.align_ms
        acc["pack_ms"] += st.pack_ms
        acc["trace_ms"] += st.trace_ms
        acc["launches"] += st.align_launches
        acc["steps"] += 1

    def reset_acc():
        for k in acc:
            acc[k] = 0

    for _ in range(args.warmup):
        step()
    reset_acc()
    elapsed = shardlib.timed_steps(step, args.steps, 0, dist=dist, sync=torch.cuda.synchronize, device="cuda")
    d_scores, ptrs = last["out"]

    st = al.stats()
    total_pairs = n_pairs * args.steps * world
    value = total_pairs / elapsed
    gcups = dptt * args.steps * world / elapsed / 1e9

    out = None
    if rank == 0:
        total_b, kernel_b = algorithmic_bytes(meta, int(st.cells), int(st.text_bytes), cigar)
        k_ms = acc["align_ms"] / max(1, acc["launches"])            # average wavefront-kernel launch
        launches_per_step = acc["launches"] / max(1, args.steps)
        achieved = (kernel_b / max(launches_per_step, 1e-9)) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic_bytes(args.workload) if not args.pairs and not args.max_error else (None, None)
        roofline = {"bound": "hbm", "kernel": "wfa_align_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(kernel_b / max(launches_per_step, 1e-9)),
                    "kernel_ms_avg": round(k_ms, 4), "launches_per_step": launches_per_step,
                    "cells_per_step": int(st.cells), "cells_per_s": round(st.cells / (acc["align_ms"] / args.steps * 1e-3), 1)
                    if acc["align_ms"] > 0 else None,
                    "note": "LDS-resident integer kernel: HBM fraction is low by construction, see DESIGN.md"}
        issue, _ = _pmc_per_launch(args.workload, ["SQ_INSTS_VALU", "SQ_INSTS_SALU"]) \
            if not args.pairs and not args.max_error else (None, None)
        if issue and k_ms > 0:
            # What this kernel really saturates: instruction issue.  A SIMD issues at most one vector and one
            # scalar instruction per 4 cycles (scratch/valu_rate.hip measures 4.04-4.17 cycles per integer
            # wave64 op; the CU's scalar unit serves its 4 SIMDs in turn): 1024 SIMDs at 2.4 GHz.
            peak = 1024 * 0.25 * 2.4e9 / 1e9
            roofline["issue"] = {"unit": "G wave-instr/s", "peak_per_pipe": round(peak, 1),
                                 "source": "SQ_INSTS_VALU / SQ_INSTS_SALU from the committed PMC pass of this command"}
            for pipe, key in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU")):
                ach = issue[key] / (k_ms * 1e-3) / 1e9
                roofline["issue"][pipe] = {"achieved": round(ach, 1), "frac": round(ach / peak, 3),
                                           "insts_per_launch": int(issue[key])}
        # LDS side of SURVEY.md section 8(d): 16 B of wavefront offsets per cell against 256 CU x 128 B/clk
        if acc["align_ms"] > 0:
            lds_ach = 16.0 * st.cells / (acc["align_ms"] / args.steps * 1e-3) / 1e9
            roofline["lds"] = {"achieved": round(lds_ach, 1), "peak": round(256 * 128 * 2.4, 1), "unit": "GB/s",
                               "frac": round(lds_ach / (256 * 128 * 2.4), 4), "bytes_per_cell": 16}
        out = {
            "metric": "alignments_per_sec", "value": round(value, 1), "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "pairs_per_gpu_per_step": n_pairs, "length": length,
                       "error": err, "penalties": "x=2,o=3,e=1", "max_error": max_error,
                       "compute_cigar": cigar, "sharding": f"batch-sharded x{world}, no collective"},
            "gcups": round(gcups, 2),
            "stage_ms_per_step": {"pack": round(acc["pack_ms"] / args.steps, 3),
                                  "align": round(acc["align_ms"] / args.steps, 3),
                                  "trace": round(acc["trace_ms"] / args.steps, 3)},
            "tier0": {"lds_bytes": int(st.lds_bytes_tier0), "blocks_per_cu": int(st.blocks_per_cu_tier0),
                      "pairs_retried": int(st.pairs_retried), "sub_batches": int(st.sub_batches)},
            "roofline": roofline,
        }
        # parity spot check outside the timed region: a sample against the oracle
        try:
            import oracle_lib
            k = min(2000, n_pairs)
            scores = d_scores[:k].cpu().numpy()
            so, co, _ = oracle_lib.oracle_batch(buf, meta[:k], PEN, cigar=cigar, nthreads=usable_cores())
            ok = bool(np.array_equal(scores, so))
            if cigar:
                cg = wfagpu.fetch_cigars(ptrs[0], ptrs[1], ptrs[2], n_pairs, st.text_bytes)[:k]
                ok = ok and cg == co
            out["parity_sample"] = {"pairs": k, "bit_exact_vs_oracle": ok}
        except Exception as ex:  # the checker is optional for the measurement itself
            out["parity_sample"] = {"error": str(ex)}
        if world == 1 and not args.no_cpu_baseline:
            # ~10-30 s of CPU work: bounded sample of the same workload
            per_pair_us = 2.0 if length <= 200 else 95.0 * (length / 1000.0) ** 2
            budget = int(max(2000, min(n_pairs, 20e6 / per_pair_us)))
            out["cpu_baseline"], _ = cpu_baseline(buf, meta, cigar, budget)
    al.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
