#!/usr/bin/env python3
"""bench.py -- aligned DP cells/s of the `sedef align` DP hot path on MI355X.

One "step" = one pass of the hot path (extz2 DP -> traceback -> CIGAR compaction -> result
gather) over one batch of synthetic DP tasks whose packed sequences are already resident in HBM.
Workload at every N: BASELINE.json configs[1] per GPU -- 100,000 tasks, query = 1000 uniform
ACGT, target = query with 6 % substitution draws / 2 % deletions / 2 % insertions, band w=128,
scoring 5/-4/40/1, zdrop=-1, flag=0.  N GPUs = N independent shards
(weak scaling); after the DP every rank all-gathers the per-task result records and CIGARs
(RCCL over xGMI), as the north star asks.

Prints ONE JSON line (rank 0).  `value` = in-band cells of all ranks / max-over-ranks time.
"""
import argparse
import os

# The chunk pipeline of the DP path keeps four streams busy (two for DP launches, traceback, the caller's).  The HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with torch's own streams);
# two of ours on one queue serialise what should overlap (measured: 27.7 instead of 22.6 ms per step).  Must be
# set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def synth_batch(n, qlen, seed, sub=0.06, dele=0.02, ins=0.02):
    """Vectorised version of the survey's mutation model. Returns (pool_codes, q_off, qlen[], t_off, tlen[])."""
    rng = np.random.Generator(np.random.MT19937(seed))
    q = rng.integers(0, 4, size=(n, qlen), dtype=np.uint8)
    r = rng.random((n, qlen))
    base = q.copy()
    m_sub = r < sub
    base[m_sub] = rng.integers(0, 4, size=int(m_sub.sum()), dtype=np.uint8)
    cnt = np.ones((n, qlen), np.int64)
    cnt[(r >= sub) & (r < sub + dele)] = 0
    m_ins = (r >= sub + dele) & (r < sub + dele + ins)
    cnt[m_ins] = 2
    empty = cnt.sum(1) == 0
    cnt[empty, 0] = 1
    flat_cnt = cnt.reshape(-1)
    t_flat = np.repeat(base.reshape(-1), flat_cnt)
    ends = np.cumsum(flat_cnt)
    ins_pos = ends[m_ins.reshape(-1)] - 1
    t_flat[ins_pos] = rng.integers(0, 4, size=len(ins_pos), dtype=np.uint8)
    tlen = cnt.sum(1).astype(np.int32)
    t_off = np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.int64)
    qlen_a = np.full(n, qlen, np.int32)
    q_off = (np.arange(n, dtype=np.int64) * qlen)
    pool = np.concatenate([q.reshape(-1), t_flat])
    return pool, q_off, qlen_a, t_off + n * qlen, tlen


def _mutated_pairs(rng, qlens, sub=0.06, dele=0.02, ins=0.02, tlens=None):
    """Per-task version of the mutation model for ragged batches: returns (pool, q_off, qlen, t_off, tlen)."""
    chunks, q_off, t_off, tl, off = [], [], [], [], 0
    for k, ql in enumerate(qlens):
        q = rng.integers(0, 4, size=int(ql), dtype=np.uint8)
        r = rng.random(int(ql))
        base = q.copy()
        m_sub = r < sub
        base[m_sub] = rng.integers(0, 4, size=int(m_sub.sum()), dtype=np.uint8)
        cnt = np.ones(int(ql), np.int64)
        cnt[(r >= sub) & (r < sub + dele)] = 0
        m_ins = (r >= sub + dele) & (r < sub + dele + ins)
        cnt[m_ins] = 2
        if cnt.sum() == 0:
            cnt[0] = 1
        t = np.repeat(base, cnt)
        t[np.cumsum(cnt)[m_ins] - 1] = rng.integers(0, 4, size=int(m_ins.sum()), dtype=np.uint8)
        if tlens is not None:  # forced target length: cut or extend with random bases
            want = int(tlens[k])
            t = t[:want] if len(t) >= want else np.concatenate([t, rng.integers(0, 4, size=want - len(t), dtype=np.uint8)])
        q_off.append(off)
        off += len(q)
        t_off.append(off)
        off += len(t)
        tl.append(len(t))
        chunks += [q, t]
    return (np.concatenate(chunks), np.array(q_off, np.int64), np.array(qlens, np.int32), np.array(t_off, np.int64),
            np.array(tl, np.int32))


def synth_hg19_mixture(n, seed, big=3000):
    """BASELINE configs[3] shape: the DP task-size mixture of captured `sedef align` task
    streams, all w=-1: ~59 % gap fills of <=100 cells, ~40 % of <=1e4 cells, ~1.2 % 500x500 side extensions,
    ~0.06 % up to 1000x1000 and ~0.06 % beyond 1e6 cells (here up to `big` x `big`).  Returns the batch and w."""
    rng = np.random.Generator(np.random.MT19937(seed))
    u = rng.random(n)
    ql = np.empty(n, np.int64)
    tl = np.empty(n, np.int64)
    for k in range(n):
        if u[k] < 0.59:
            a, b = int(rng.integers(1, 11)), int(rng.integers(1, 11))
        elif u[k] < 0.99:
            a = int(rng.integers(5, 101))
            b = max(1, min(209, a + int(rng.integers(-20, 21))))
        elif u[k] < 0.9988:
            a, b = 500, int(rng.choice([500, 500, 500, 431, 377]))
        elif u[k] < 0.9994:
            a, b = int(rng.integers(600, 1001)), int(rng.integers(600, 1001))
        else:
            a = int(rng.integers(1200, big + 1))
            b = a
        ql[k], tl[k] = a, b
    return _mutated_pairs(rng, ql, tlens=tl), -1


def synth_mm8_mixture(n, seed, max_len=20000):
    """BASELINE configs[4] shape: lengths log-uniform 200..max_len, one band of
    {64,128,256,512} per task, 5 % of the tasks with a one-sided 1-5 kb indel.  Returns the batch and w[]."""
    rng = np.random.Generator(np.random.MT19937(seed))
    ql = np.exp(rng.uniform(np.log(200), np.log(max_len), n)).astype(np.int64)
    w = rng.choice([64, 128, 256, 512], size=n).astype(np.int32)
    batch = list(_mutated_pairs(rng, ql))
    pool, q_off, qlen, t_off, tlen = batch
    # one-sided indel: cut 1-5 kb out of the middle of the target of every 20th task (where it is long enough)
    chunks, new_t_off, new_tlen, off = [], [], [], 0
    for k in range(n):
        q = pool[q_off[k]:q_off[k] + qlen[k]]
        t = pool[t_off[k]:t_off[k] + tlen[k]]
        if k % 20 == 0 and len(t) > 6000:
            cut = int(rng.integers(1000, 5001))
            at = int(rng.integers(0, len(t) - cut))
            t = np.concatenate([t[:at], t[at + cut:]])
        chunks += [q, t]
        new_t_off.append(off + len(q))
        new_tlen.append(len(t))
        off += len(q) + len(t)
    q_off2 = np.array(new_t_off, np.int64) - qlen
    return (np.concatenate(chunks), q_off2, qlen, np.array(new_t_off, np.int64), np.array(new_tlen, np.int32)), w


def pack_batch(pool, q_off, qlen, t_off, tlen):
    """Packs every sequence (2-bit codes + N mask) into one uint32 pool; returns (words, q_word, t_word)."""
    import sedef_amd
    lib = sedef_amd.load_library()
    n = len(qlen)
    qw = np.array([sedef_amd.packed_words(int(x)) for x in np.unique(qlen)])
    qmap = dict(zip(np.unique(qlen).tolist(), qw.tolist()))
    tuniq = np.unique(tlen)
    tmap = dict(zip(tuniq.tolist(), [sedef_amd.packed_words(int(x)) for x in tuniq]))
    q_words = np.array([qmap[int(x)] for x in qlen], np.int64)
    t_words = np.array([tmap[int(x)] for x in tlen], np.int64)
    offs = np.zeros(2 * n + 1, np.int64)
    inter = np.empty(2 * n, np.int64)
    inter[0::2], inter[1::2] = q_words, t_words
    offs[1:] = np.cumsum(inter)
    words = np.zeros(int(offs[-1]), np.uint32)
    base = words.ctypes.data
    pbase = pool.ctypes.data
    for k in range(n):
        lib.sdf_pack_codes(pbase + int(q_off[k]), int(qlen[k]), base + 4 * int(offs[2 * k]))
        lib.sdf_pack_codes(pbase + int(t_off[k]), int(tlen[k]), base + 4 * int(offs[2 * k + 1]))
    return words, offs[0:2 * n:2].copy(), offs[1:2 * n:2].copy()


def algorithmic_bytes(qlen, tlen, cells, n_cigar):
    """2-bit inputs + N mask + 1 B/cell direction write + traceback read bound +
    result record, per task."""
    s = qlen.astype(np.int64) + tlen
    return int(((s + 3) // 4 + (s + 7) // 8 + cells + s + 4 * n_cigar.astype(np.int64) + 64).sum())


def cpu_baseline(pool, q_off, qlen, t_off, tlen, cells_task, w, budget_s=10.0):
    """Times the CPU path on a bounded sample of the same workload on this box's host cores: one
    task stream per hardware thread, each a C loop over ksw_extz2_sse calls (the reference runs
    one single-threaded process per bucket file, reference: sedef.sh:187-190).
    kind = "reference": the reference kernel itself (oracle/_ref, compiled from /root/reference in
    the build container); "port": our scalar oracle, if that build is absent."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.binding import Oracle, Reference
    try:
        impl, kind = Reference(), "reference"
    except Exception:
        impl, kind = Oracle(), "port"
    cores = effective_cores()
    n = len(qlen)
    chunk = 32
    deadline = [0.0]

    def stream(i):  # one task stream: C loops over `chunk` tasks until the wall-clock budget is spent
        cells, k = 0, (i * 7919 * chunk) % n
        while time.perf_counter() < deadline[0]:
            idx = np.arange(k, k + chunk) % n
            impl.batch(pool, q_off[idx], qlen[idx], t_off[idx], tlen[idx], w=w)
            cells += int(cells_task[idx].sum())
            k = (k + chunk) % n
        return cells

    t0 = time.perf_counter()
    deadline[0] = t0 + budget_s
    with ThreadPoolExecutor(cores) as ex:
        per = list(ex.map(stream, range(cores)))
    dt = time.perf_counter() - t0
    cells = sum(per)
    return {"value": round(cells / dt / 1e9, 4), "unit": "Gcell/s", "cores": cores, "kind": kind,
            "sample": "%d tasks drawn from the same batch, %d concurrent single-threaded streams "
                      "(host exposes %d logical CPUs), %.1f s wall" % (cells // max(int(cells_task.mean()), 1),
                                                                        cores, os.cpu_count() or 1, dt)}


def effective_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tasks", type=int, default=100000, help="DP tasks per GPU per step")
    ap.add_argument("--qlen", type=int, default=1000)
    ap.add_argument("--band", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workspace-gib", type=float, default=48.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import sedef_amd
    from sedef_amd.dist import ResultGather, allgatherv_results

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback of the product path)")
    # BENCH_DEBUG_ONE_GPU=1: dry-run of the multi-rank path on a single-GPU box (all ranks on
    # device 0, gloo instead of RCCL).  Never used for reported numbers.
    debug_one_gpu = os.environ.get("BENCH_DEBUG_ONE_GPU") == "1"
    if debug_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if debug_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n, w = args.tasks, args.band
    pool, q_off, qlen, t_off, tlen = synth_batch(n, args.qlen, seed=42 + rank)
    words, q_word, t_word = pack_batch(pool, q_off, qlen, t_off, tlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
    tasks["w"], tasks["zdrop"], tasks["flag"] = w, -1, 0
    cells = np.array([sedef_amd.band_cells(int(a), int(b), w) for a, b in
                      zip(*np.unique(np.stack([qlen, tlen]), axis=1))])
    keys = {(int(a), int(b)): int(c) for (a, b), c in
            zip(np.unique(np.stack([qlen, tlen]), axis=1).T, cells)}
    cells_task = np.array([keys[(int(a), int(b))] for a, b in zip(qlen, tlen)], np.int64)
    cells_rank = int(cells_task.sum())

    eng = sedef_amd.Extz2Engine(local, int(args.workspace_gib * (1 << 30)))
    d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
    cig_cap = int((qlen.astype(np.int64) + tlen + 2).sum())
    cig_cap = min(cig_cap, 256 * n)  # ~50 runs per task at 10 % divergence; overflow is an error
    # N > 1: two sets of output buffers in rotation -- the all-gather of step i's records and CIGAR words (RCCL, on the
    # communicator's stream) runs under the DP of step i+1, which writes the other set; every gather is waited for
    # inside the timed region.  The CIGAR words travel with a fixed capacity of 64 per task (~48 are used).
    nsets = 2 if world > 1 else 1
    d_outs = [torch.empty(n * 16, dtype=torch.int32, device=dev) for _ in range(nsets)]
    d_cigs = [torch.empty(cig_cap, dtype=torch.int32, device=dev) for _ in range(nsets)]
    d_out, d_cig = d_outs[0], d_cigs[0]
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    stream = torch.cuda.current_stream().cuda_stream

    gathers = []
    if world > 1:
        gdev = torch.device("cpu") if debug_one_gpu else dev
        gathers = [ResultGather(n * 16, min(64 * n, cig_cap), gdev, torch.int32) for _ in range(nsets)]
    step_no = [0]
    sync_gathered = [None, None, None]

    def step():
        b = step_no[0] % nsets
        step_no[0] += 1
        if gathers:
            gathers[b].wait()  # the gather that last read this buffer set
        used = eng.align_batch_device(tasks, d_pool.data_ptr(), d_outs[b].data_ptr(), d_cigs[b].data_ptr(),
                                      cig_cap, want=want, stream=stream)
        if gathers:  # all-gather of result records + CIGAR words over RCCL, asynchronous
            try:
                if debug_one_gpu:
                    gathers[b].start(d_outs[b].cpu(), d_cigs[b].cpu(), used)
                else:
                    gathers[b].start(d_outs[b], d_cigs[b], used)
            except (RuntimeError, ValueError) as e:  # (same on every rank) -> the synchronous gather from here on
                if rank == 0:
                    print("asynchronous gather unavailable (%s): synchronous all-gather" % e, file=sys.stderr)
                del gathers[:]
        if world > 1 and not gathers:
            src = (d_outs[b].cpu(), d_cigs[b].cpu()) if debug_one_gpu else (d_outs[b], d_cigs[b])
            sync_gathered[:] = allgatherv_results(src[0], src[1], used)
        return used

    def sync():
        for g in gathers:
            g.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    dp_ms, tb_ms, cp_ms, launches, plan_ms, call_ms = 0.0, 0.0, 0.0, 0, 0.0, 0.0
    t0 = time.perf_counter()
    used = 0
    for _ in range(args.steps):
        used = step()
        dp_ms += eng.last_ms(0)
        tb_ms += eng.last_ms(1)
        cp_ms += eng.last_ms(2)
        plan_ms += eng.last_ms(4)
        call_ms += eng.last_ms(5)
        launches += eng.last_launches()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        rdev = torch.device("cpu") if debug_one_gpu else dev
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([cells_rank], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot)
        cells_all = int(tot.item())
    else:
        cells_all = cells_rank

    if gathers:  # every rank holds every rank's results: check this rank's own slice of the last gather
        lb = (step_no[0] - 1) % nsets
        ra, ca, cnts = gathers[lb].result()
        assert int(cnts[rank, 0]) == n * 16 and int(cnts[rank, 1]) == used
        assert torch.equal(ra[rank].cpu(), d_outs[lb].cpu()) and torch.equal(ca[rank][:used].cpu(), d_cigs[lb][:used].cpu())
    if rank == 0:
        res = d_out.cpu().numpy().view(sedef_amd.RESULT_DTYPE)
        assert int(res["n_cigar"].astype(np.int64).sum()) == used
        alg = algorithmic_bytes(qlen, tlen, cells_task, res["n_cigar"])
        # Roofline of the dominant kernel.  The timed steps above run the batch as a pipeline of chunks whose DP
        # launches overlap each other and the traceback, so a launch's own duration is taken from one extra,
        # untimed pass of the same batch through a context created with SDF_PIPELINE=0: the whole batch in ONE
        # DP launch on one stream, HIP events around it on that stream (sdf_last_ms(0)).  The committed rocprofv3
        # summary (profiles/) is of `SDF_PIPELINE=0 python bench.py ...`, the same launch.
        iso_ms, iso_launches = dp_ms / max(args.steps, 1), max(launches // max(args.steps, 1), 1)
        roof_mode = "pipelined launches (union of the DP intervals)"
        if world == 1:
            old = os.environ.get("SDF_PIPELINE")
            os.environ["SDF_PIPELINE"] = "0"
            try:
                iso = sedef_amd.Extz2Engine(local, int(args.workspace_gib * (1 << 30)))
            finally:
                if old is None:
                    del os.environ["SDF_PIPELINE"]
                else:
                    os.environ["SDF_PIPELINE"] = old
            for _ in range(2):  # warm-up + measured
                iso.align_batch_device(tasks, d_pool.data_ptr(), d_out.data_ptr(), d_cig.data_ptr(), cig_cap,
                                       want=want, stream=stream)
            iso_ms, iso_launches = iso.last_ms(0), max(iso.last_launches(), 1)
            roof_mode = "one isolated DP launch of the whole batch (SDF_PIPELINE=0 pass, untimed)"
            del iso
        avg_launch_s = iso_ms / 1e3 / iso_launches
        bytes_per_launch = alg / iso_launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        valu = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp) and n == 100000 and w == 128 and args.qlen == 1000:
            # measured once with rocprofv3 PMC passes on this exact workload and launch (profiles/README.md)
            pmc = json.load(open(tp))
            traffic = pmc.get("bytes_per_step") / iso_launches
            if pmc.get("valu_insts_per_step"):
                # what actually bounds the launch: VALU issue.  A SIMD issues one wavefront VALU instruction per 4
                # cycles at best (64 lanes over 16 ALUs); counted instructions / (SIMDs x launch cycles / 4)
                props = torch.cuda.get_device_properties(local)
                simds = props.multi_processor_count * 4
                mhz = (getattr(props, "clock_rate", 0) or 2400000) / 1e3  # (2.4 GHz: MI355X peak engine clock)
                issue_peak = simds * mhz * 1e6 / 4.0 * avg_launch_s * iso_launches
                valu = {"insts_per_step": pmc["valu_insts_per_step"], "simds": simds, "clock_mhz": round(mhz, 1),
                        "frac_of_issue_peak": round(pmc["valu_insts_per_step"] / issue_peak, 4),
                        "source": pmc.get("valu_source")}
        value = cells_all * args.steps / dt / 1e9
        line = {
            "metric": "aligned DP cells/sec (Gcell/s) on `sedef align` batch",
            "value": round(value, 3), "unit": "Gcell/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic %dx~%d DP tasks per GPU, band=%d, "
                                   "affine gap 5/-4/40/1, CIGAR+score+counts" % (n, args.qlen, args.qlen, w),
                       "tasks_per_gpu": n, "band": w, "cells_per_step_per_gpu": cells_rank,
                       "parallelism": "task-sharded x%d + all-gatherv of result records" % world},
            "kernel_ms_per_step": {"pipeline_chunks": launches / args.steps,
                                   "dp": round(dp_ms / args.steps, 3), "traceback": round(tb_ms / args.steps, 3),
                                   "compact": round(cp_ms / args.steps, 3),
                                   "host_planning": round(plan_ms / args.steps, 3),
                                   "host_call_total": round(call_ms / args.steps, 3)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "extz2_pair_kernel<3> (extz2 DP)", "measured_on": roof_mode,
                         "launches": iso_launches, "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch), "valu_issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pool, q_off, qlen, t_off, tlen, cells_task, w)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
