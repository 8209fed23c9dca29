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
