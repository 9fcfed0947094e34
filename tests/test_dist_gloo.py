"""CPU, world_size 2 over gloo: task sharding, the all-gatherv of result records, and the strong-scaling flow of
bench.py (shard -> align -> gatherv -> union check) end to end with a stand-in for the engine."""
import os
import socket
import time

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, *args, world=2):
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(fn, args=(world, port, out) + args, nprocs=world, join=True)
    assert all(out.get(r) is True for r in range(world)), dict(out)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _fake_results(ids):
    """Stand-in for the engine: per task 16 int32 (score = 3 * id, n_cigar = id % 7, cigar_off) + its CIGAR words."""
    rec = np.zeros((len(ids), 16), np.int32)
    rec[:, 0] = 3 * ids
    ncig = (ids % 7).astype(np.int64)
    rec[:, 9] = ncig
    off = np.cumsum(ncig) - ncig
    rec[:, 10:12] = off.astype(np.int64).view(np.int32).reshape(-1, 2)
    cig = np.concatenate([np.arange(n, dtype=np.int32) + 16 * t for t, n in zip(ids, ncig)] + [np.zeros(0, np.int32)])
    return rec, cig


def test_shard_is_a_balanced_partition_and_fast():
    from sedef_amd.dist import shard_tasks
    rng = np.random.default_rng(1)
    cost = np.concatenate([rng.integers(1, 100, 5000), rng.integers(10 ** 5, 10 ** 8, 20)])
    for world in (1, 2, 4, 8):
        sh = shard_tasks(cost, world)
        assert np.array_equal(np.sort(np.concatenate(sh)), np.arange(len(cost)))
        loads = np.array([cost[s].sum() for s in sh], dtype=np.float64)
        assert loads.max() <= loads.mean() + cost.max()
    # the hg19-shaped mixture at the north star's batch size: 1,000,000 tasks, eight ranks, well under 50 ms each
    u = rng.random(1000000)
    cost = np.where(u < 0.59, rng.integers(1, 100, len(u)), np.where(u < 0.99, rng.integers(25, 10000, len(u)),
                    np.where(u < 0.9994, 250000, rng.integers(1200, 6000, len(u)) ** 2)))
    t0 = time.perf_counter()
    sh = shard_tasks(cost, 8)
    dt = time.perf_counter() - t0
    loads = np.array([cost[s].sum() for s in sh], dtype=np.float64)
    assert np.array_equal(np.sort(np.concatenate(sh)), np.arange(len(cost)))
    assert loads.max() / loads.mean() < 1.02 and dt < 2.0, (loads.max() / loads.mean(), dt)  # (0.3-0.5 s on a quiet 8-CPU container)


def _worker_gatherv(rank, world, port, out):
    import torch
    dist = _init(rank, world, port)
    from sedef_amd.dist import allgatherv_results, shard_tasks
    try:
        rng = np.random.default_rng(5)
        cost = rng.integers(1, 10 ** 6, size=1001)
        shards = shard_tasks(cost, world)
        rec, cig = _fake_results(shards[rank])
        pad = np.concatenate([cig, np.full(13, -1, np.int32)])  # pool larger than `used`
        ra, ca, counts = allgatherv_results(torch.from_numpy(rec.reshape(-1)), torch.from_numpy(pad), len(cig))
        ok = ra.numel() == 16 * len(cost) and int(counts[:, 1].sum()) == ca.numel()  # exact sizes, no padding
        ro = co = 0
        for r in range(world):
            er, ec = _fake_results(shards[r])
            ok &= int(counts[r, 0]) == er.size and int(counts[r, 1]) == len(ec)
            ok &= np.array_equal(ra[ro:ro + er.size].numpy(), er.reshape(-1)) and np.array_equal(ca[co:co + len(ec)].numpy(), ec)
            ro, co = ro + er.size, co + len(ec)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_allgatherv_world2_gloo():
    _spawn(_worker_gatherv)


def _worker_gatherv_ragged(rank, world, port, out):
    """Three ranks, one with records but no CIGAR words, one with nothing at all: the grouped point-to-point transfers
    skip the empty ranges on both sides and still match."""
    import torch
    dist = _init(rank, world, port)
    from sedef_amd.dist import ResultGatherV
    try:
        ids = [np.arange(0, 700, 7), np.arange(1, 400), np.zeros(0, np.int64)]  # rank 0: n_cigar = id % 7 = 0 everywhere
        g = ResultGatherV(torch.device("cpu"), torch.int32)
        ok = True
        for step in range(3):  # (buffers are reused: same group of transfers again)
            rec, cig = _fake_results(ids[rank])
            g.start(torch.from_numpy(rec.reshape(-1)), torch.from_numpy(cig), len(cig))
            ra, ca, counts = g.result()
            for r in range(world):
                er, ec = _fake_results(ids[r])
                pr, pc = g.part(r)
                ok &= np.array_equal(pr.numpy(), er.reshape(-1)) and np.array_equal(pc.numpy(), ec)
            ok &= counts[:, 0].tolist() == [1600, 399 * 16, 0] and int(counts[0, 1]) == 0 and int(counts[2, 1]) == 0
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_allgatherv_world3_empty_ranges_gloo():
    _spawn(_worker_gatherv_ragged, world=3)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher typed by the caller: the script starts N ranks itself, before it
    touches the GPU.  Without a GPU (this test runs in the build container) every rank must refuse to run -- the product
    path has no CPU fallback -- and the launcher must hand the failure on; under a launcher whose world size differs from
    --gpus it refuses as well."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_bench_launch.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a HIP device") == 2, p.stderr  # one refusal per rank
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE is 2" in p.stderr


def _worker_strong(rank, world, port, out):
    """bench.py --strong in miniature: every rank aligns its shard step after step into two buffer sets in rotation, the
    gatherv of step i runs while step i+1 'computes', and the union of the shards equals the whole batch."""
    import torch
    dist = _init(rank, world, port)
    from sedef_amd.dist import ResultGatherV, shard_tasks, task_checksums
    try:
        rng = np.random.default_rng(9)
        n = 3000
        cost = np.where(rng.random(n) < 0.99, rng.integers(1, 5000, n), rng.integers(10 ** 6, 10 ** 7, n))
        shards = shard_tasks(cost, world)
        mine = shards[rank]
        whole_rec, whole_cig = _fake_results(np.arange(n))
        whole = task_checksums(whole_rec, whole_cig)
        sets = [ResultGatherV(torch.device("cpu"), torch.int32) for _ in range(2)]
        ok = True
        for step in range(4):
            b = step % 2
            sets[b].wait()  # the gather that last read this buffer set
            rec, cig = _fake_results(mine)
            sets[b].start(torch.from_numpy(rec.reshape(-1)), torch.from_numpy(cig), len(cig))
        for g in sets:
            g.wait()
        union = np.zeros(n, np.uint64)
        seen = np.zeros(n, np.int32)
        for r in range(world):
            pr, pc = sets[1].part(r)
            union[shards[r]] = task_checksums(pr.numpy(), pc.numpy())
            seen[shards[r]] += 1
        ok &= bool((seen == 1).all()) and np.array_equal(union, whole)
        try:
            sets[0].start(torch.zeros(16, dtype=torch.int32), torch.zeros(4, dtype=torch.int32), 5)
            ok = False
        except ValueError:
            pass
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_strong_scaling_flow_world2_gloo():
    _spawn(_worker_strong)


def test_task_checksums_see_every_word():
    from sedef_amd.dist import task_checksums
    rec, cig = _fake_results(np.arange(50))
    base = task_checksums(rec, cig)
    assert len(np.unique(base)) == 50
    c2 = cig.copy()
    c2[len(c2) // 2] ^= 1
    assert (task_checksums(rec, c2) != base).sum() == 1
    r2 = rec.copy()
    r2[7, 0] += 1
    assert (task_checksums(r2, cig) != base).sum() == 1
    # a task's checksum does not depend on where its words sit in the pool
    perm = np.arange(50)[::-1]
    rec_p, cig_p = _fake_results(perm)
    assert np.array_equal(task_checksums(rec_p, cig_p)[::-1], base)
