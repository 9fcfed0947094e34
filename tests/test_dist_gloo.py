"""CPU, world_size 2 over gloo: task sharding and the all-gatherv of result records."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    from sedef_amd.dist import allgatherv_results, shard_tasks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        cost = rng.integers(1, 10 ** 6, size=1001)
        shards = shard_tasks(cost, world)
        mine = shards[rank]
        # fake per-task results: 16 int32 per record, record k tagged with its global task id,
        # and a CIGAR pool of (task id % 7) words per task
        rec = np.zeros((len(mine), 16), np.int32)
        rec[:, 0] = mine
        ncig = mine % 7
        rec[:, 9] = ncig
        cig = np.concatenate([np.full(n, t, np.int32) for t, n in zip(mine, ncig)] + [np.zeros(0, np.int32)])
        pad = np.concatenate([cig, np.full(13, -1, np.int32)])  # pool larger than `used`
        ra, ca, counts = allgatherv_results(torch.from_numpy(rec.reshape(-1)), torch.from_numpy(pad), len(cig))
        ok = True
        seen = []
        for r in range(world):
            n = int(counts[r, 0]) // 16
            recs = ra[r][: n * 16].view(n, 16).numpy()
            ok &= np.array_equal(recs[:, 0], shards[r])
            words = ca[r][: int(counts[r, 1])].numpy()
            exp = np.concatenate([np.full(t % 7, t, np.int32) for t in shards[r]] + [np.zeros(0, np.int32)])
            ok &= np.array_equal(words, exp)
            seen.append(recs[:, 0])
        ok &= np.array_equal(np.sort(np.concatenate(seen)), np.arange(len(cost)))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_shard_is_a_balanced_partition():
    from sedef_amd.dist import shard_tasks
    rng = np.random.default_rng(1)
    cost = np.concatenate([rng.integers(1, 100, 5000), rng.integers(10 ** 5, 10 ** 8, 20)])
    for world in (1, 2, 4, 8):
        sh = shard_tasks(cost, world)
        allidx = np.sort(np.concatenate(sh))
        assert np.array_equal(allidx, np.arange(len(cost)))
        loads = np.array([cost[s].sum() for s in sh], dtype=np.float64)
        assert loads.max() <= loads.mean() + cost.max()


def test_allgatherv_world2_gloo():
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out.get(0) is True and out.get(1) is True


def _worker_async(rank, world, port, out):
    import torch
    import torch.distributed as dist

    from sedef_amd.dist import ResultGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, cap = 50, 400
        sets = [ResultGather(n * 16, cap, torch.device("cpu"), torch.int32) for _ in range(2)]
        ok = True
        expect = {}
        for step in range(5):  # two buffer sets in rotation: the gather of step i is read after step i+1 started
            b = step % 2
            rec = torch.full((n * 16,), 1000 * step + rank, dtype=torch.int32)
            used = 100 + 7 * rank + step
            cig = torch.full((cap + 9,), -1, dtype=torch.int32)
            cig[:used] = 10 * step + rank
            sets[b].start(rec, cig, used)
            expect[b] = step
            if step:
                pb = (step - 1) % 2
                ra, ca, counts = sets[pb].result()
                for r in range(world):
                    ps = expect[pb]
                    ok &= bool((ra[r] == 1000 * ps + r).all()) and int(counts[r, 0]) == n * 16
                    u = int(counts[r, 1])
                    ok &= u == 100 + 7 * r + ps and bool((ca[r][:u] == 10 * ps + r).all())
        try:
            sets[0].start(torch.zeros(n * 16, dtype=torch.int32), torch.zeros(cap, dtype=torch.int32), cap + 1)
            ok = False
        except ValueError:
            pass
        for s_ in sets:
            s_.wait()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_async_result_gather_world2_gloo():
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_async, args=(2, port, out), nprocs=2, join=True)
    assert out.get(0) is True and out.get(1) is True
