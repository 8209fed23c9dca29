"""The N>1 path of bench.py on CPU: two gloo ranks shard a batch, time steps with the barrier/max-over-ranks
protocol, and rank 0 aggregates.  The 'step' here is the ORACLE (a checker), because the product has no CPU
path; what is under test is the sharding and timing harness, not the kernels."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import shardlib


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            cuts = [shardlib.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, os.path.join(root, "wfa-gpu_amd", "bindings"))
    import time
    import oracle_lib
    import shardlib as sl
    import wfagpu
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = sl.init_distributed("gloo")
    buf, meta = wfagpu.generate_pairs(64, 120, 0.05, seed=sl.shard_seed(500, rank))
    state = {}

    def step():
        if rank == 1:
            time.sleep(0.05)   # the slow rank must set the reported time
        state["scores"], _, _ = oracle_lib.oracle_batch(buf, meta, (2, 3, 1), cigar=False)

    elapsed = sl.timed_steps(step, steps=2, warmup=1, dist=dist)
    q.put((rank, elapsed, int(state["scores"].sum()), len(meta)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_harness():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, s0, n0), (r1, e1, s1, n1) = res
    assert e0 == e1 >= 0.1               # max over ranks, identical on both
    assert s0 != s1                      # different shards (different seeds)
    assert n0 + n1 == 128                # whole-job units = sum over ranks


def _bench_json(args, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout      # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment must start 2 ranks itself (a torch.distributed.run child,
    spawned before the parent touches any GPU) and report n_gpus = the ranks that really ran.  --cpu-harness swaps the
    GPU step for the oracle and RCCL for gloo so that this launch path is covered here."""
    out = _bench_json(["--gpus", "2", "--cpu-harness", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["harness_only"] is True
    assert out["library_call"] == {"ranks_parked": 1}      # the rank-0-alone leg ran between two host barriers
    # every rank's own clock next to the aggregate (which uses the slowest)
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["ms_per_step"] <= out["ms_per_step"] * 1.001 for r in out["per_rank"])
    one = _bench_json(["--cpu-harness", "--steps", "2", "--warmup", "1"])
    assert one["n_gpus"] == 1
    # --force-dist: ONE rank, but as a torch.distributed job all the same (self-spawned child, process group, barriers, all-reduce
    # and all-gather of the clocks): what the GPU suite runs with RCCL on the one device it has
    forced = _bench_json(["--gpus", "1", "--force-dist", "--cpu-harness", "--steps", "2", "--warmup", "1"])
    assert forced["n_gpus"] == 1 and forced["library_call"] == {"ranks_parked": 0} and len(forced["per_rank"]) == 1


def test_bench_eight_rank_harness():
    """The launch path of the driver's 8-GPU run (self-spawned torchrun child, eight ranks, barriers, max-over-ranks, all-gather of the
    clocks, rank-0-alone leg between two host barriers) with gloo and the oracle as the step; every rank's generator takes its share of
    the host's cores, not sixteen threads apiece (eight ranks x sixteen threads started at once before the clock)."""
    out = _bench_json(["--gpus", "8", "--cpu-harness", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 8 and out["harness_only"] is True
    assert [r["rank"] for r in out["per_rank"]] == list(range(8))
    assert out["library_call"] == {"ranks_parked": 7}
    assert all(r["ms_per_step"] <= out["ms_per_step"] * 1.001 for r in out["per_rank"])
    cores = len(os.sched_getaffinity(0))
    assert 1 <= out["generator_threads_per_rank"] <= max(1, cores // 8)


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """Launched by torchrun with 2 ranks but --gpus 4: the line would misreport n_gpus, so the run must fail."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--cpu-harness", "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "started 2 ranks" in r.stderr + r.stdout
