"""Multi-GPU plumbing of the DP hot path: one process per GPU, task sharding, result gather.

DP tasks are independent (the reference kernel is re-entrant and `sedef align` is run as
independent single-threaded processes, reference: sedef.sh:187-190), so the batch is sharded
across ranks without any data-path collective.  The one exchange step is the all-gatherv of the
per-task result records and CIGAR words after the DP (RCCL over xGMI when the backend is nccl):
payload ~0.3 kB per task, latency-bound, one hop on the fully connected mesh.
"""
import heapq

import numpy as np

REC_WORDS = 16  # int32 words of one sdf_result record (include/sedef_hip.h)


def shard_tasks(cost, world, head=None):
    """Partition of tasks by DP cells: longest-processing-time-first for the heavy head, then the tail (cheap tasks,
    the bulk of a batch) is cut into contiguous runs of its cost-sorted order that fill every rank up to the mean.

    cost: per-task cell counts.  Returns a list of `world` index arrays (each sorted ascending) with near-equal total
    cost -- within the cost of one head task of the mean; every task appears in exactly one shard.  O(n log n) in
    numpy plus O(head log world) in Python (head = 2048 tasks per rank unless given)."""
    cost = np.asarray(cost, dtype=np.int64)
    n = len(cost)
    if world <= 1:
        return [np.arange(n)]
    order = np.argsort(-cost, kind="stable")
    nh = min(n, 2048 * world if head is None else int(head))
    owner = np.empty(n, np.int32)
    load = [(0, r) for r in range(world)]
    heapq.heapify(load)
    for i in order[:nh].tolist():  # exact LPT on the tasks that can unbalance a shard
        l, r = heapq.heappop(load)
        owner[i] = r
        heapq.heappush(load, (l + int(cost[i]), r))
    loads = np.zeros(world, np.int64)
    for l, r in load:
        loads[r] = l
    tail = order[nh:]
    if len(tail):
        # every rank is filled up to the common mean: rank r takes the tail tasks whose running cost falls into its
        # share [cum_need[r-1], cum_need[r]); ranks already above the mean take none
        tc = cost[tail]
        total = loads.sum() + tc.sum()
        need = np.maximum(total / world - loads, 0.0)
        need *= tc.sum() / max(need.sum(), 1.0)
        edges = np.cumsum(need)
        run = np.cumsum(tc) - tc  # cost of the tail before each task
        owner[tail] = np.minimum(np.searchsorted(edges, run, side="right"), world - 1).astype(np.int32)
    return [np.flatnonzero(owner == r) for r in range(world)]


def allgatherv_results(records, cigars, used, group=None):
    """Blocking all-gatherv of the results of every rank (see ResultGatherV).  Returns (records_all, cigars_all,
    counts [world, 2] = (record words, CIGAR words) per rank); rank r's part starts at counts[:r].sum(0)."""
    g = ResultGatherV(records.device, records.dtype, group=group)
    g.start(records, cigars, used)
    return g.result()


class ResultGatherV:
    """All-gatherv of one step's result records and CIGAR words: every rank ends with every rank's results, back to
    back in rank order, exactly `counts[r]` words from rank r (no padding travels).

    `start` exchanges the counts (one small all-gather and a host read of its 2 * world words: the exact ranges size the
    receive buffers), then enqueues ONE group of point-to-point transfers on the exact sizes -- RCCL has no native
    gatherv; two collectives per step whatever the world size -- and returns; the caller computes the next step into ANOTHER set of
    buffers and calls `wait` before it reuses this set or reads `result`.  With the nccl backend `wait` makes the
    CURRENT torch stream wait for the collectives: the caller must launch the work that reuses the buffers on that
    stream (bench.py runs the engine on a torch stream for this reason).  Buffers grow on demand and are reused."""

    def __init__(self, device, dtype, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device, self.dtype = device, dtype
        self.mine = torch.zeros(2, dtype=torch.int64, device=device)
        self.counts_dev = torch.zeros(self.world * 2, dtype=torch.int64, device=device)
        self.recs = torch.empty(0, dtype=dtype, device=device)
        self.cig = torch.empty(0, dtype=dtype, device=device)
        self.counts = None
        self.handles = []
        self.keep = None
        # the counts come to the host through pinned memory behind an event: the host waits for that copy alone, not for
        # the device (a `.cpu()` of the table synchronises the whole current stream)
        self.on_gpu = torch.device(device).type == "cuda"
        self.counts_host = torch.zeros(self.world * 2, dtype=torch.int64)
        if self.on_gpu:
            self.counts_host = self.counts_host.pin_memory()
            self.counts_ev = torch.cuda.Event()
        # loopback (world == 1 only, bench.py's BENCH_FORCE_DIST=1): this rank's two ranges also travel to ITSELF through
        # the communicator's send / receive pair, into buffers of their own -- the point-to-point path of the all-gatherv
        # on a box with one GPU
        self.loopback = False
        self.loop_recs = self.loop_cig = None

    def reserve(self, rec_words, cig_words):
        """Sizes the receive buffers once (all ranks' records and CIGAR words): no step allocates after this."""
        import torch
        if self.recs.numel() < rec_words:
            self.recs = torch.empty(int(rec_words), dtype=self.dtype, device=self.device)
        if self.cig.numel() < cig_words:
            self.cig = torch.empty(int(cig_words), dtype=self.dtype, device=self.device)

    def start(self, records, cigars, used):
        import torch
        import torch.distributed as dist
        used = int(used)
        if used > cigars.numel():
            raise ValueError("ResultGatherV: %d CIGAR words used, %d in the buffer" % (used, cigars.numel()))
        self.wait()
        self.mine[0] = records.numel()
        self.mine[1] = used
        dist.all_gather_into_tensor(self.counts_dev, self.mine, group=self.group)
        if self.on_gpu:
            self.counts_host.copy_(self.counts_dev, non_blocking=True)
            self.counts_ev.record()
            self.counts_ev.synchronize()
        else:
            self.counts_host.copy_(self.counts_dev)
        counts = self.counts_host.view(self.world, 2).clone()
        self.counts = counts
        rec_off = np.concatenate([[0], np.cumsum(counts[:, 0].numpy())])
        cig_off = np.concatenate([[0], np.cumsum(counts[:, 1].numpy())])
        if self.recs.numel() < rec_off[-1]:
            self.recs = torch.empty(int(rec_off[-1] * 1.25) + 16, dtype=self.dtype, device=self.device)
        if self.cig.numel() < cig_off[-1]:
            self.cig = torch.empty(int(cig_off[-1] * 1.25) + 16, dtype=self.dtype, device=self.device)
        self.rec_off, self.cig_off = rec_off, cig_off
        # own part: a local copy; the other parts: broadcasts from their owners into the exact ranges
        self.recs[rec_off[self.rank]:rec_off[self.rank + 1]].copy_(records)
        self.cig[cig_off[self.rank]:cig_off[self.rank + 1]].copy_(cigars[:used])
        self.keep = (records, cigars)  # the sources must stay alive until the transfers are done
        # ONE group of point-to-point transfers (batch_isend_irecv: ncclGroupStart ... ncclGroupEnd on RCCL): this rank's
        # two ranges to every peer, every peer's two ranges into their places here.  Every rank derives the same list
        # from the same counts table, so sends and receives of a pair match in order; empty ranges are skipped on both
        # sides.  On the fully connected xGMI mesh the world - 1 transfers of a rank run side by side, one hop each.
        ops = []
        me = self.rank
        for r in range(self.world):
            if r == me:
                continue
            peer = dist.get_global_rank(self.group, r) if self.group is not None else r
            for buf, off in ((self.recs, rec_off), (self.cig, cig_off)):
                if off[me + 1] > off[me]:
                    ops.append(dist.P2POp(dist.isend, buf[off[me]:off[me + 1]], peer, self.group))
                if off[r + 1] > off[r]:
                    ops.append(dist.P2POp(dist.irecv, buf[off[r]:off[r + 1]], peer, self.group))
        if self.loopback and self.world == 1:
            if self.loop_recs is None or self.loop_recs.numel() < rec_off[1] or self.loop_cig.numel() < cig_off[1]:
                self.loop_recs, self.loop_cig = torch.empty_like(self.recs), torch.empty_like(self.cig)
            peer = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            for buf, dst, off in ((self.recs, self.loop_recs, rec_off), (self.cig, self.loop_cig, cig_off)):
                if off[1] > off[0]:
                    ops.append(dist.P2POp(dist.isend, buf[off[0]:off[1]], peer, self.group))
                    ops.append(dist.P2POp(dist.irecv, dst[off[0]:off[1]], peer, self.group))
        self.handles = dist.batch_isend_irecv(ops) if ops else []

    def loopback_part(self):
        """(records, CIGAR words) as they arrived through the loopback transfers of the last `start`."""
        self.wait()
        return self.loop_recs[:int(self.rec_off[1])], self.loop_cig[:int(self.cig_off[1])]

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        self.keep = None

    def result(self):
        """(records of all ranks back to back, CIGAR words back to back, counts [world, 2]) of the last `start`."""
        self.wait()
        return self.recs[:int(self.rec_off[-1])], self.cig[:int(self.cig_off[-1])], self.counts

    def part(self, r):
        """Rank r's (records, CIGAR words) of the last `start`."""
        self.wait()
        return (self.recs[int(self.rec_off[r]):int(self.rec_off[r + 1])],
                self.cig[int(self.cig_off[r]):int(self.cig_off[r + 1])])


def task_checksums(records, cigars):
    """One 64-bit checksum per task over (score, n_cigar, CIGAR words): records = int32 array [n, 16] in sdf_result
    layout (cigar_off = words 10..11, little endian), cigars = the rank's CIGAR pool.  Order independent per task, so
    the union of the shards can be compared with a single-GPU run of the whole batch."""
    rec = np.ascontiguousarray(records, dtype=np.int32).reshape(-1, REC_WORDS)
    cig = np.ascontiguousarray(cigars, dtype=np.int32).view(np.uint32).astype(np.uint64)
    n = len(rec)
    ncig = rec[:, 9].astype(np.int64)
    off = rec[:, 10:12].copy().view(np.int64).reshape(-1)
    h = (rec[:, 0].astype(np.int64).view(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ ncig.view(np.uint64)
    if n and ncig.sum():
        # position-weighted sum of every task's words: word j of the task times an odd multiplier of j + 1
        idx = np.repeat(off, ncig) + (np.arange(int(ncig.sum())) - np.repeat(np.cumsum(ncig) - ncig, ncig))
        j = (np.arange(int(ncig.sum())) - np.repeat(np.cumsum(ncig) - ncig, ncig)).astype(np.uint64)
        w = cig[idx] * (np.uint64(2) * j + np.uint64(0x100000001B3))
        acc = np.zeros(n, np.uint64)
        np.add.at(acc, np.repeat(np.arange(n), ncig), w)
        h ^= acc * np.uint64(0xC2B2AE3D27D4EB4F)
    return h
