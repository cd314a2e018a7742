"""
Utterance sharding across the GPUs of one node.

The reference's only parallelism is one utterance per multiprocessing.Pool worker with no exchange of data
(libutils.py:32-63).  The MI355X counterpart keeps that model: one process per GPU (torch.distributed over
RCCL only for rendezvous, barriers and gathering a few scalars), utterances dealt to ranks
longest-processing-time first, NO data-path collective -- frames of different utterances never interact.
"""
import heapq
import os

import numpy as np


def shard_by_cost(costs, world_size):
    """
    LPT assignment: returns a list (one entry per rank) of index arrays into ``costs``; every index appears
    exactly once; ranks' total costs differ by at most one item's cost.
    """
    costs = np.asarray(costs, dtype=np.float64)
    world_size = int(world_size)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    heap = [(0.0, r) for r in range(world_size)]
    out = [[] for _ in range(world_size)]
    for i in np.argsort(-costs, kind="stable"):
        load, r = heapq.heappop(heap)
        out[r].append(int(i))
        heapq.heappush(heap, (load + float(costs[i]), r))
    return [np.asarray(sorted(x), dtype=np.int64) for x in out]


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def local_device_index():
    """GPU of this rank: LOCAL_RANK, or 0 for every rank when MAGPHASE_SHARE_DEVICE=1 (exercising the N > 1 code paths
    on a box with one GPU: tests, bench.py's BENCH_SHARE_DEVICE)."""
    if os.environ.get("MAGPHASE_SHARE_DEVICE") or os.environ.get("BENCH_SHARE_DEVICE"):
        return 0
    return dist_env()[1]


def init_process_group(backend=None):
    """Rendezvous on 127.0.0.1 (the container hostname may not resolve).  backend: 'nccl' (= RCCL) on GPUs, 'gloo' on CPU."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = dist_env()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_device_index())
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_device_index()))
    else:
        dist.init_process_group(backend=backend)
    return dist


def run_sharded(process_shard, costs, gather_scalars=True):
    """
    Runs ``process_shard(indices) -> dict of python scalars`` on this rank's share of the items and (optionally)
    gathers the per-rank scalar dicts on every rank.  The only communication is that gather of a few numbers.
    """
    import torch.distributed as dist

    rank, _, world = dist_env()
    shards = shard_by_cost(costs, world)
    mine = process_shard(shards[rank])
    if world == 1 or not gather_scalars:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out
