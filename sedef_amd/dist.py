"""Multi-GPU plumbing of the DP hot path: one process per GPU, task sharding, result gather.

DP tasks are independent (the reference kernel is re-entrant and `sedef align` is run as
independent single-threaded processes, reference: sedef.sh:187-190), so the batch is sharded
across ranks without any data-path collective.  The one exchange step is the all-gatherv of the
per-task result records and CIGAR words after the DP (RCCL over xGMI when the backend is nccl):
payload ~0.3 kB per task, latency-bound, one hop on the fully connected mesh.
"""
import numpy as np


def shard_tasks(cost, world):
    """Longest-processing-time-first partition of tasks by DP cells.

    cost: per-task cell counts.  Returns a list of `world` index arrays (each sorted ascending)
    with near-equal total cost; every task appears in exactly one shard."""
    cost = np.asarray(cost, dtype=np.int64)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world, np.int64)
    owner = np.empty(len(cost), np.int32)
    # chunked greedy: exact LPT for the heavy head, round-robin by current load for the rest
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cost[i]
    return [np.flatnonzero(owner == r) for r in range(world)]


def allgatherv_results(records, cigars, used, group=None):
    """All-gather the variable-length results of every rank.

    records: int32 tensor [n*16] (sdf_result records of this rank), cigars: int32 tensor holding
    `used` CIGAR words (the rest is padding).  Returns (records_all [world, n_max*16],
    cigars_all [world, c_max], counts [world, 2] = (n_records*16, cigar_words)) on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = records.device
    mine = torch.tensor([records.numel(), int(used)], dtype=torch.int64, device=dev)
    counts = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = counts.view(world, 2)
    n_max, c_max = int(counts[:, 0].max().item()), max(int(counts[:, 1].max().item()), 1)
    rec_pad = records if records.numel() == n_max else \
        torch.cat([records, records.new_zeros(n_max - records.numel())])
    cig = cigars[:min(int(used), cigars.numel())]
    cig_pad = cig if cig.numel() == c_max else torch.cat([cig, cig.new_zeros(c_max - cig.numel())])
    rec_all = torch.empty(world * n_max, dtype=records.dtype, device=dev)
    cig_all = torch.empty(world * c_max, dtype=cigars.dtype, device=dev)
    dist.all_gather_into_tensor(rec_all, rec_pad.contiguous(), group=group)
    dist.all_gather_into_tensor(cig_all, cig_pad.contiguous(), group=group)
    return rec_all.view(world, n_max), cig_all.view(world, c_max), counts
