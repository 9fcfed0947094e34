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


class ResultGather:
    """All-gather of one step's results into preallocated buffers, asynchronously: `start` enqueues the three
    collectives (counts, result records, CIGAR words up to a fixed capacity) and returns; the caller computes the
    next step into ANOTHER set of buffers and calls `wait` before it reuses this set (or reads `result`).  Over nccl
    (RCCL) the collectives run on the communicator's stream under the next step's kernels; nothing blocks the host.

    Every rank passes the same `rec_words` (records are fixed-size) and `cig_cap` (words gathered per rank: the
    valid prefix is `counts[r, 1]`, the rest is padding)."""

    def __init__(self, rec_words, cig_cap, device, dtype, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rec_words, self.cig_cap = int(rec_words), int(cig_cap)
        self.mine = torch.zeros(2, dtype=torch.int64, device=device)
        self.counts = torch.zeros(self.world * 2, dtype=torch.int64, device=device)
        self.recs = torch.empty(self.world * self.rec_words, dtype=dtype, device=device)
        self.cig = torch.empty(self.world * self.cig_cap, dtype=dtype, device=device)
        self.handles = []

    def start(self, records, cigars, used):
        import torch.distributed as dist
        if records.numel() != self.rec_words or int(used) > self.cig_cap or cigars.numel() < self.cig_cap:
            raise ValueError("ResultGather: %d record words / %d CIGAR words do not fit (%d / %d)" %
                             (records.numel(), int(used), self.rec_words, self.cig_cap))
        self.wait()
        self.mine[0] = records.numel()
        self.mine[1] = int(used)
        self.handles = [
            dist.all_gather_into_tensor(self.counts, self.mine, group=self.group, async_op=True),
            dist.all_gather_into_tensor(self.recs, records, group=self.group, async_op=True),
            dist.all_gather_into_tensor(self.cig, cigars[:self.cig_cap], group=self.group, async_op=True),
        ]

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []

    def result(self):
        """(records [world, rec_words], cigars [world, cig_cap], counts [world, 2]) of the last `start`."""
        self.wait()
        return (self.recs.view(self.world, self.rec_words), self.cig.view(self.world, self.cig_cap),
                self.counts.view(self.world, 2))
