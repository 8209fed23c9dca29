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
    # (one node: the host group of rank0_exclusive needs no name lookup -- only when the rendezvous itself is on loopback:
    # a multi-node launch must keep gloo on a routable interface)
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, **kw)
    return dist


def timed_steps(step, steps, warmup, dist=None, sync=lambda: None, device="cpu", per_rank=False):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides; returns the MAX
    over ranks of the elapsed seconds -- with per_rank, (that, [every rank's own elapsed seconds up to its device sync])."""
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
    own = time.perf_counter() - t0        # this rank's K steps, before it waits for the others
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks = [own]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if per_rank:
            mine = torch.tensor([own], dtype=torch.float64, device=device)
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            ranks = [float(v.item()) for v in every]
    return (elapsed, ranks) if per_rank else elapsed


def host_group(dist):
    """A second process group on gloo (CPU sockets) for waits that must not occupy the GPUs: a barrier of the RCCL group is
    a kernel that spins on every rank's device.  Collective: every rank calls it, right after init_distributed."""
    return dist.new_group(backend="gloo")


def rank0_exclusive(dist, group, rank, fn):
    """Every rank meets at a host barrier, rank 0 alone runs fn() (it may use every GPU of the node: the other ranks are
    parked in a socket wait, their devices idle), every rank meets again.  Returns fn()'s result on rank 0, None elsewhere."""
    dist.barrier(group=group)
    out = None
    err = None
    if rank == 0:
        try:
            out = fn()
        except Exception as ex:     # the other ranks must not be left waiting
            err = ex
    dist.barrier(group=group)
    if err is not None:
        raise err
    return out
