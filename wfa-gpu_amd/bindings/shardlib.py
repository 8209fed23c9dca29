"""Batch sharding helpers shared by bench.py and the tests: pairs are independent, so N ranks (one per GPU)
each take a contiguous slice / their own seeded shard; the only cross-rank traffic is a barrier and the
max-over-ranks of the wall clock."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n pairs for `rank` -- the same cut launch_alignments* makes per device
    (csrc/wfa_launch.hip: from = n*d/ndev, to = n*(d+1)/ndev)."""
    return n * rank // world, n * (rank + 1) // world


def shard_seed(base_seed, rank):
    return base_seed + rank


def init_distributed(backend, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, **kw)
    return dist


def timed_steps(step, steps, warmup, dist=None, sync=lambda: None, device="cpu"):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed seconds."""
    import time
    import torch
    for _ in range(warmup):
        step()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
