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


# ----------------------------------------------------------------------------------------------------------------------
# Host side of a multi-rank job: every rank on its own cores, next to its GPU.  The reference's model is one worker per
# core with nothing shared (libutils.py:61-62: Pool() = one process per core); eight ranks that each raise 32 staging
# threads on whatever cores the scheduler picks share one memory system and migrate across NUMA nodes instead.
# ----------------------------------------------------------------------------------------------------------------------
def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index):
    """NUMA node of a GPU from sysfs (its PCI function's numa_node), or None when the platform does not say."""
    try:
        import torch

        p = torch.cuda.get_device_properties(int(device_index))
        bdf = "%04x:%02x:%02x.0" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as fh:
            node = int(fh.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def rank_core_sets(device_of_rank, allowed, node_of_device=gpu_numa_node, cpus_of_node=None):
    """
    The cores of every local rank: the cores of its GPU's NUMA node (those in `allowed`) split evenly, in rank order, among
    the ranks whose GPUs sit on that node; ranks whose node is unknown (or has no allowed core) split what is left over
    evenly.  device_of_rank: device index per local rank.  Every rank gets at least one core; the sets are disjoint whenever
    there are at least as many cores as ranks.  Pure function of its arguments (tests pass the two lookups).
    """
    if cpus_of_node is None:
        def cpus_of_node(n):
            try:
                with open("/sys/devices/system/node/node%d/cpulist" % n) as fh:
                    return _parse_cpulist(fh.read())
            except OSError:
                return []
    allowed = sorted(set(int(c) for c in allowed))
    W = len(device_of_rank)
    by_node, loose = {}, []
    for r, d in enumerate(device_of_rank):
        n = node_of_device(d)
        cores = [c for c in cpus_of_node(n) if c in set(allowed)] if n is not None else []
        if cores:
            by_node.setdefault(n, (cores, []))[1].append(r)
        else:
            loose.append(r)
    out = [None] * W
    taken = set()
    for n, (cores, ranks) in by_node.items():
        if len(cores) < len(ranks):      # fewer cores than ranks on this node: they share the node's cores
            for r in ranks:
                out[r] = list(cores)
            taken.update(cores)
            continue
        per = len(cores) // len(ranks)
        for k, r in enumerate(ranks):
            out[r] = cores[k * per:(k + 1) * per]
            taken.update(out[r])
    if loose:
        rest = [c for c in allowed if c not in taken] or allowed
        per = max(1, len(rest) // len(loose))
        for k, r in enumerate(loose):
            out[r] = rest[(k * per) % len(rest):(k * per) % len(rest) + per] or rest[:1]
    return out


def bind_rank_to_cores(local_rank=None, local_world=None, shared_device=None):
    """
    os.sched_setaffinity of this process (and of every native thread it starts afterwards: the staging pools of the host
    helpers inherit it) to this rank's share of the cores next to its GPU.  Call once, early, in every rank of a node.
    Returns {"cores": n, "numa_node": node or None, "first_core": c} for the records, or None when nothing was changed
    (one rank, MAGPHASE_BIND_CORES=0, no sched_setaffinity on this platform).
    """
    if os.environ.get("MAGPHASE_BIND_CORES", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    _, lr, world = dist_env()
    local_rank = lr if local_rank is None else int(local_rank)
    local_world = int(local_world if local_world is not None else os.environ.get("LOCAL_WORLD_SIZE", world))
    if local_world <= 1:
        return None
    if shared_device is None:
        shared_device = bool(os.environ.get("MAGPHASE_SHARE_DEVICE") or os.environ.get("BENCH_SHARE_DEVICE"))
    devices = [0 if shared_device else r for r in range(local_world)]
    try:
        allowed = os.sched_getaffinity(0)
        sets = rank_core_sets(devices, allowed)
        mine = sets[local_rank]
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        return {"cores": len(mine), "numa_node": gpu_numa_node(devices[local_rank]), "first_core": int(mine[0])}
    except OSError:
        return None
