#!/usr/bin/env python3
"""bench.py -- aligned DP cells/s of the `sedef align` DP hot path on MI355X.

One "step" = one pass of the hot path (planning -> extz2 DP -> traceback -> CIGAR compaction -> result D2H, and at
N > 1 the all-gatherv of the result records and CIGARs over RCCL / xGMI) over one batch of synthetic DP tasks whose
packed sequences are already resident in HBM.

Default workload, at every N: BASELINE.json configs[1] per GPU -- 100,000 tasks, query = 1000 uniform ACGT, target =
query with 6 % substitution draws / 2 % deletions / 2 % insertions, band w=128, scoring 5/-4/40/1, zdrop=-1, flag=0
(SURVEY.md 8(d), config 2); N GPUs = N independent batches (weak scaling).
`--workload hg19mix --tasks 1000000 --strong`: the north star's batch -- ONE hg19-shaped mixture of a million tasks
(BASELINE configs[3]) sharded over the ranks by cells (sedef_amd.dist.shard_tasks), every rank aligns its shard, every
rank receives every rank's results, and rank 0 checks the union against a single-GPU run of the whole batch.

Prints ONE JSON line (rank 0).  `value` = in-band cells of all ranks / max-over-ranks time; `value_incl_pcie` adds the
H2D of the packed sequences (untimed extra passes, single GPU).
"""
import argparse
import os

# The chunk pipeline of the DP path keeps four streams busy (two for DP launches, traceback, the caller's).  The HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with torch's own streams);
# two of ours on one queue serialise what should overlap (measured: 27.7 instead of 22.6 ms per step).  Must be
# set before the runtime initialises.  Sixteen since two batch calls are in flight on two contexts (--inflight 2: 1,167-1,174 ->
# 1,168-1,186 Gcell/s on configs[1], the hg19 mixture 1,047-1,087 -> 1,101-1,112, same box, alternating; eight with one call).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
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


def synth_ragged(rng, qlens, tlens=None, sub=0.06, dele=0.02, ins=0.02):
    """Vectorised _mutated_pairs for ragged batches of any size (same mutation model, one pass over all bases):
    returns (pool, q_off, qlen, t_off, tlen).  With `tlens` every target is cut, or extended with random bases, to the
    given length."""
    qlens = np.asarray(qlens, np.int64)
    n = len(qlens)
    q_start = np.cumsum(qlens) - qlens
    total = int(qlens.sum())
    q = rng.integers(0, 4, size=total, dtype=np.uint8)
    r = rng.random(total)
    base = q.copy()
    m_sub = r < sub
    base[m_sub] = rng.integers(0, 4, size=int(m_sub.sum()), dtype=np.uint8)
    cnt = np.ones(total, np.int64)
    cnt[(r >= sub) & (r < sub + dele)] = 0
    m_ins = (r >= sub + dele) & (r < sub + dele + ins)
    cnt[m_ins] = 2
    t_all = np.repeat(base, cnt)
    t_all[np.cumsum(cnt)[m_ins] - 1] = rng.integers(0, 4, size=int(m_ins.sum()), dtype=np.uint8)
    nat = np.add.reduceat(cnt, q_start) if n else np.zeros(0, np.int64)
    nat_start = np.cumsum(nat) - nat
    want = np.maximum(nat, 1) if tlens is None else np.asarray(tlens, np.int64)
    # target k = its first min(nat, want) mutated bases, then random bases up to `want`
    t_start = np.cumsum(want) - want
    j = np.arange(int(want.sum())) - np.repeat(t_start, want)
    src = np.repeat(nat_start, want) + j
    have = j < np.repeat(nat, want)
    t = rng.integers(0, 4, size=len(j), dtype=np.uint8)
    t[have] = t_all[src[have]]
    pool = np.concatenate([q, t])
    return pool, q_start, qlens.astype(np.int32), t_start + total, want.astype(np.int32)


def synth_hg19_mixture_fast(n, seed, big=6000):
    """synth_hg19_mixture's size mixture (BASELINE configs[3]: ~59 % of <= 100 cells, ~40 % of <= 1e4, ~1.2 % 500 x 500
    side extensions, ~0.06 % up to 1000 x 1000, ~0.06 % up to `big` x `big`, all w = -1) drawn and mutated in vectorised
    form: 1,000,000 tasks in a few seconds.  Returns the batch and w."""
    rng = np.random.Generator(np.random.MT19937(seed))
    u = rng.random(n)
    a1, b1 = rng.integers(1, 11, n), rng.integers(1, 11, n)
    a2 = rng.integers(5, 101, n)
    b2 = np.clip(a2 + rng.integers(-20, 21, n), 1, 209)
    b3 = rng.choice(np.array([500, 500, 500, 431, 377]), n)
    a4, b4 = rng.integers(600, 1001, n), rng.integers(600, 1001, n)
    a5 = rng.integers(1200, big + 1, n)
    ql = np.select([u < 0.59, u < 0.99, u < 0.9988, u < 0.9994], [a1, a2, np.full(n, 500), a4], a5)
    tl = np.select([u < 0.59, u < 0.99, u < 0.9988, u < 0.9994], [b1, b2, b3, b4], a5)
    return synth_ragged(rng, ql, tlens=tl), -1


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


def synth_mm8_mixture_fast(n, seed, max_len=20000, slab=8192):
    """synth_mm8_mixture's distribution (BASELINE configs[4]: lengths log-uniform 200..max_len, one band of {64, 128, 256,
    512} per task, every 20th task with a target of more than 6000 bases loses 1-5 kb from its middle) drawn and mutated
    in vectorised form, slab by slab: 100,000 tasks (~0.9 G bases) in well under a minute.  Returns the batch and w[]."""
    rng = np.random.Generator(np.random.MT19937(seed))
    ql_all = np.exp(rng.uniform(np.log(200), np.log(max_len), n)).astype(np.int64)
    w = rng.choice([64, 128, 256, 512], size=n).astype(np.int32)
    pools, q_off, t_off, tlens, base = [], [], [], [], 0
    for s0 in range(0, n, slab):
        ql = ql_all[s0:s0 + slab]
        pool, qs, _, ts, tl = synth_ragged(rng, ql)
        tl = tl.astype(np.int64)
        idx = np.arange(s0, s0 + len(ql))
        cut_it = (idx % 20 == 0) & (tl > 6000)
        cut = np.where(cut_it, rng.integers(1000, 5001, len(ql)), 0)
        at = (rng.random(len(ql)) * np.maximum(tl - cut, 1)).astype(np.int64)
        # keep mask over the slab's target bases: position j of task k stays unless at <= j < at + cut
        j = np.arange(int(tl.sum())) - np.repeat(np.cumsum(tl) - tl, tl)
        keep = ~((j >= np.repeat(at, tl)) & (j < np.repeat(at + cut, tl)))
        total_q = int(ql.sum())
        t_new = pool[total_q:][keep]
        tl2 = tl - cut
        pools += [pool[:total_q], t_new]
        q_off.append(qs + base)
        t_off.append(np.cumsum(tl2) - tl2 + base + total_q)
        tlens.append(tl2)
        base += total_q + len(t_new)
    return (np.concatenate(pools), np.concatenate(q_off).astype(np.int64), ql_all.astype(np.int32),
            np.concatenate(t_off).astype(np.int64), np.concatenate(tlens).astype(np.int32)), w


def pack_batch(pool, q_off, qlen, t_off, tlen):
    """Packs every sequence (2-bit codes + N mask) into one uint32 pool; returns (words, q_word, t_word)
    (include/sedef_hip.h: sdf_pack_tasks)."""
    import ctypes as C

    import sedef_amd
    lib = sedef_amd.load_library()
    n = len(qlen)
    pool = np.ascontiguousarray(pool, np.uint8)
    q_off, t_off = np.ascontiguousarray(q_off, np.int64), np.ascontiguousarray(t_off, np.int64)
    qlen, tlen = np.ascontiguousarray(qlen, np.int32), np.ascontiguousarray(tlen, np.int32)
    q_word, t_word = np.zeros(n, np.int64), np.zeros(n, np.int64)
    lib.sdf_pack_tasks.restype = C.c_size_t
    lib.sdf_pack_tasks.argtypes = [C.c_void_p] * 5 + [C.c_size_t] + [C.c_void_p] * 3
    total = lib.sdf_pack_tasks(pool.ctypes.data, q_off.ctypes.data, qlen.ctypes.data, t_off.ctypes.data, tlen.ctypes.data, n,
                               None, q_word.ctypes.data, t_word.ctypes.data)
    words = np.zeros(int(total), np.uint32)
    lib.sdf_pack_tasks(pool.ctypes.data, q_off.ctypes.data, qlen.ctypes.data, t_off.ctypes.data, tlen.ctypes.data, n,
                       words.ctypes.data, q_word.ctypes.data, t_word.ctypes.data)
    return words, q_word, t_word


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
            "per_core": round(cells / dt / 1e9 / cores, 4), "host_cores_total": os.cpu_count() or 1,
            "sample": "%d tasks drawn from the same batch, %d concurrent single-threaded streams "
                      "(host exposes %d logical CPUs), %.1f s wall" % (cells // max(int(cells_task.mean()), 1),
                                                                        cores, os.cpu_count() or 1, dt)}


def spot_check(pool, q_off, qlen, t_off, tlen, w, res, cigs, n_sample=64, seed=7):
    """In-run check, outside the timed region: `n_sample` random tasks of the LAST timed step -- score and every CIGAR word
    as they came back over PCIe -- against the CPU checker on the same sequences (the reference kernel itself when
    oracle/_ref is built, else the scalar oracle).  Raises on the first difference."""
    from oracle.binding import Oracle, Reference
    try:
        impl, kind = Reference(), "reference"
    except Exception:
        impl, kind = Oracle(), "port"
    rng = np.random.default_rng(seed)
    n = len(qlen)
    idx = rng.choice(n, size=min(n_sample, n), replace=False)
    ws = np.broadcast_to(np.asarray(w, np.int64), (n,))
    for k in idx.tolist():
        q = pool[q_off[k]:q_off[k] + qlen[k]]
        t = pool[t_off[k]:t_off[k] + tlen[k]]
        exp = impl.extz2(q, t, w=int(ws[k]))
        r = res[k]
        got = cigs[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        if int(r["score"]) != int(exp["score"]) or not np.array_equal(np.asarray(got, np.uint32), np.asarray(exp["cigar"], np.uint32)):
            raise AssertionError("spot check: task %d (%d x %d, w=%d) differs from the %s kernel" % (k, qlen[k], tlen[k], ws[k], kind))
    return {"tasks": int(len(idx)), "checker": kind, "compared": "score + every CIGAR word of the last timed step", "ok": True}



def stage_block(budget_s=120.0):
    """The STAGE the north star calls the drop-in (`sedef align generate`, reference: src/align_main.cc:285-337), untimed
    extra after the headline: the product CLI in processes of its own on BASELINE configs[2]'s workload at size
    (tests/hostgen.py: make_chr1_genome -- a 249 Mb chromosome, ~1,860 seed pairs of 1-100 kb, fwd + rc).
      parity     the four buckets of tests/test_stage_scale.py in ONE process; sha256 of their BEDPE in bucket order against
                 the committed tests/golden/stage_chr1.sha256 (= this repository's host code on the reference kernel)
      one bucket all seed pairs in one bucket file, one process: the stage's own clock (the reference prints the same
                 "Finished BED ... in Xs" line, src/align_main.cc:335-336) and the process's wall time
      8 buckets  the whole seed file under eight names, one process (sedef.sh:187-190 starts one per bucket): the clock
                 around all eight stages
      cpu        the same host code with the REFERENCE kernel as the DP on the usable host cores (oracle/_ref behind the
                 library's test hook): one chr1-sized bucket -- the stage-level CPU figure next to the GPU's
    Never raises: a failure is reported in the block."""
    import hashlib
    import re
    import shutil
    import subprocess
    import tempfile
    t_start = time.time()
    out = {"what": "`sedef align generate` (product CLI, own processes) on the chr1-sized synthetic genome of BASELINE configs[2]; "
                   "untimed extra, not part of `value`"}
    d = tempfile.mkdtemp(prefix="sdf_bench_stage_")
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostgen
        from sedef_amd import host
        if not (os.path.exists(host.CLI) and os.path.exists(host.LIB)):
            raise RuntimeError("host CLI not built (run __graft_entry__.build())")
        fa = os.path.join(d, "genome.fa")
        genome, nseeds = hostgen.make_chr1_genome(fa)
        out["genome_bp"] = sum(len(v) for v in genome.values())
        del genome
        for name, nb in (("four", 4), ("one", 1)):
            os.makedirs(os.path.join(d, name))
            host.bucket(fa + ".seeds.bed", nb, os.path.join(d, name), fa)
        os.makedirs(os.path.join(d, "eight"))
        for k in range(8):
            shutil.copy(os.path.join(d, "one", "bucket_0000"), os.path.join(d, "eight", "bucket_%04d" % k))
        out["seed_pairs"] = nseeds

        def cli(args, stdout=None):
            t0 = time.perf_counter()
            r = subprocess.run([host.CLI, "align", "generate", "-k", "11", fa] + args, stdout=stdout, stderr=subprocess.PIPE,
                               timeout=max(20.0, budget_s - (time.time() - t_start)))
            err = r.stderr.decode(errors="replace").replace("\r", "\n")
            if r.returncode != 0:
                raise RuntimeError("sedef align generate failed (%d): %s" % (r.returncode, err[-300:]))
            return time.perf_counter() - t0, err

        # parity: four buckets, one process, the committed hash
        cli([os.path.join(d, "four")])
        text = b"".join(open(os.path.join(d, "four", "bucket_%04d.aligned.bed" % k), "rb").read() for k in range(4))
        sha = hashlib.sha256(text).hexdigest()
        committed = open(os.path.join(ROOT, "tests", "golden", "stage_chr1.sha256")).read().split()[0]
        out["parity"] = {"lines": text.count(b"\n"), "sha256": sha, "committed_sha256": committed, "ok": sha == committed,
                         "checker": "tests/golden/stage_chr1.sha256: the same host code on the reference's ksw_extz2_sse"}
        # one chr1-sized bucket, one process (three runs: best and all)
        clocks, walls = [], []
        for _ in range(3):
            with open(os.path.join(d, "one.bed"), "wb") as f:
                wall, err = cli([os.path.join(d, "one", "bucket_0000")], stdout=f)
            clocks.append(float(re.search(r"Finished BED \S+ in ([0-9.]+)s", err).group(1)))
            walls.append(round(wall, 3))
        tasks = re.search(r"(\d+) tasks, ([0-9.e+]+) cells", err)
        out["one_bucket"] = {"stage_clock_s": min(clocks), "stage_clock_s_runs": clocks, "process_wall_s_runs": walls,
                             "lines": open(os.path.join(d, "one.bed"), "rb").read().count(b"\n"),
                             "dp_tasks": int(tasks.group(1)) if tasks else None,
                             "dp_cells": float(tasks.group(2)) if tasks else None}
        # eight chr1-sized buckets, one process
        runs = []
        for _ in range(2):
            for fn in os.listdir(os.path.join(d, "eight")):
                if fn.endswith(".aligned.bed"):
                    os.remove(os.path.join(d, "eight", fn))
            wall, err = cli([os.path.join(d, "eight")], stdout=subprocess.DEVNULL)
            m = re.search(r"All 8 buckets done in ([0-9.]+)s \((\d+) lines", err)
            runs.append({"all_stages_s": float(m.group(1)), "process_wall_s": round(wall, 3), "lines": int(m.group(2)),
                         "stage_clocks_s": [float(x) for x in re.findall(r"Finished BED \S+ in ([0-9.]+)s", err)]})
        same = all(open(os.path.join(d, "eight", "bucket_%04d.aligned.bed" % k), "rb").read() ==
                   open(os.path.join(d, "one.bed"), "rb").read() for k in range(8))
        out["eight_buckets_one_process"] = {"all_stages_s": min(r["all_stages_s"] for r in runs), "runs": runs,
                                            "every_bucket_equals_the_single_run": same}
        # the CPU leg: same host code, reference kernel, usable cores
        try:
            import ctypes as C
            from oracle.binding import build_reference
            so = build_reference()
            lib = C.CDLL(so)
            hook = C.cast(lib.ref_extz2_hook, C.c_void_p)
            t0 = time.perf_counter()
            st = host.generate(fa, os.path.join(d, "one", "bucket_0000"), 11, os.path.join(d, "cpu.bed"), test_dp=hook)
            cpu_s = time.perf_counter() - t0
            out["cpu_stage"] = {"wall_s": round(cpu_s, 3), "cores": effective_cores(), "kind": "reference kernel behind this "
                                "repository's host pipeline (the reference's own driver needs Boost: unbuildable here)",
                                "lines": open(os.path.join(d, "cpu.bed"), "rb").read().count(b"\n"), "input_lines": st[0],
                                "equals_gpu_output": open(os.path.join(d, "cpu.bed"), "rb").read() ==
                                open(os.path.join(d, "one.bed"), "rb").read(),
                                "gpu_x": round(cpu_s / min(clocks), 2)}
        except Exception as e:  # noqa: BLE001  (no reference build on this box: the GPU figures stand alone)
            out["cpu_stage"] = {"error": str(e)[:200]}
    except Exception as e:  # noqa: BLE001
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    out["seconds_spent"] = round(time.time() - t_start, 1)
    return out

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


def batch_cells(qlen, tlen, w):
    """In-band cells per task (the metric's unit; include/sedef_hip.h: sdf_band_cells)."""
    import sedef_amd
    if np.ndim(w) == 0 and int(w) < 0:
        return qlen.astype(np.int64) * tlen.astype(np.int64)
    ws = np.broadcast_to(np.asarray(w, np.int64), qlen.shape)
    key = np.stack([qlen.astype(np.int64), tlen.astype(np.int64), ws])
    uniq, inv = np.unique(key, axis=1, return_inverse=True)
    cu = np.array([sedef_amd.band_cells(int(a), int(b), int(c)) for a, b, c in uniq.T], np.int64)
    return cu[inv.reshape(-1)]


def measured_copy_bandwidth(dev, mib=1024, reps=5):
    """Achievable HBM bandwidth of this box, GB/s: a device-to-device copy of `mib` MiB timed with events on the current
    stream, bytes read + bytes written over the best of `reps` (the second roofline denominator next to the
    nominal 8 TB/s)."""
    import torch
    n = mib << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def source_sha256(rel):
    import hashlib
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def encoding_split():
    """Per-encoding instruction counts of the steady row of the dominant kernel, from the committed ISA summary
    (profiles/pair_kernel_isa.json, made by profiles/isa_split.py from the device assembly of the built library)."""
    p = os.path.join(ROOT, "profiles", "pair_kernel_isa.json")
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    row = d.get("steady_row") or {}
    return {"valu_per_row_of_two_tasks": row.get("valu"), "by_encoding": row.get("by_encoding"),
            "by_measured_issue_class": row.get("issue_class"),  # (profiles/r05_ubench_valu_ops.txt: ~2.3 and ~4.2 cycles)
            "source": "committed profile: profiles/pair_kernel_isa.json (profiles/isa_split.py on the device assembly)"}


def visible_devices():
    """HIP devices this process would see, counted WITHOUT initialising the GPU runtime (the launcher below must stay
    clean of it): the KFD topology's GPU nodes, filtered like the runtime filters them."""
    nodes = []
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in sorted(os.listdir(base), key=lambda x: int(x) if x.isdigit() else 1 << 30):
            try:
                props = dict(l.split(None, 1) for l in open(os.path.join(base, d, "properties")).read().splitlines() if " " in l)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:  # (CPU nodes have none)
                nodes.append(d)
    except OSError:
        return None
    n = len(nodes)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n):
    """Starts `n` ranks of this script (one per GPU, LOCAL_RANK = device) on this node and waits for them: the
    rendezvous variables are what `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would set."""
    import socket
    import subprocess
    have = visible_devices()
    if os.environ.get("BENCH_DEBUG_ONE_GPU") != "1" and have is not None and have < n:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, have), file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.05)
            bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if bad and not rc:  # a rank failed: the others would wait for it in a collective forever
                rc = bad[0]
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
        rc = rc or max(abs(p.returncode) for p in procs)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        rc = 130
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["configs1", "hg19mix"], default="configs1",
                    help="configs1: BASELINE configs[1], the metric's headline (default); hg19mix: the hg19-shaped task "
                         "mixture of BASELINE configs[3]")
    ap.add_argument("--tasks", type=int, default=None, help="DP tasks per step: per GPU (weak scaling, default) or in the "
                                                            "whole batch (--strong); default 100000 / 1000000 by workload")
    ap.add_argument("--strong", action="store_true", help="one batch of --tasks tasks sharded over the ranks by cells "
                                                          "(shard_tasks) instead of one batch per rank")
    ap.add_argument("--qlen", type=int, default=1000)
    ap.add_argument("--band", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie-pass", action="store_true", help="skip the untimed pass that measures value_incl_pcie")
    ap.add_argument("--no-stage", action="store_true", help="skip the untimed `stage` block (the product CLI on the chr1-sized "
                                                            "genome: parity hash, stage clocks)")
    ap.add_argument("--workspace-gib", type=float, default=48.0)
    ap.add_argument("--inflight", type=int, default=2,
                    help="batch calls in flight, each on a device context and host thread of its own (1: one call after the other)")
    args = ap.parse_args()
    args.inflight = max(1, min(3, args.inflight))
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: this process starts the N ranks itself -- before anything here
        # has touched the GPU (no torch import, no HIP call so far: a process that has initialised the GPU must not
        # spawn-and-replace) -- and exits with their worst return code.  Rank 0 prints the JSON line.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    import sedef_amd
    from sedef_amd.dist import ResultGatherV, shard_tasks, task_checksums

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run with --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback of the product path)")
    # BENCH_DEBUG_ONE_GPU=1: dry-run of the multi-rank path on a single-GPU box (all ranks on
    # device 0, gloo instead of RCCL).  Never used for reported numbers.
    debug_one_gpu = os.environ.get("BENCH_DEBUG_ONE_GPU") == "1"
    if debug_one_gpu:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d, %d visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # BENCH_FORCE_DIST=1 (with --gpus 1): the distributed path on ONE rank -- the nccl (= RCCL) backend initialised on the
    # device, every step through ResultGatherV on device tensors (count all-gather on RCCL, stream ordering through wait(),
    # the rank's own part compared) and, with BENCH_LOOPBACK unset or 1, the rank's two ranges sent to itself through the
    # communicator's point-to-point pair: what a box with one GPU can execute of the N > 1 path.
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            import socket
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if debug_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus
        # (a second, gloo group for the LAST barrier: while rank 0 times the CPU baseline the other ranks wait in a socket
        # read -- a barrier on the GPU would have them poll their streams on the host cores the baseline is measured on)
        quiet = dist.new_group(backend="gloo") if world > 1 and not debug_one_gpu else None

    # ---- the workload: every rank's shard of the batch ----
    hg19 = args.workload == "hg19mix"
    n_arg = args.tasks if args.tasks else (1000000 if hg19 else 100000)
    n_batch = n_arg if args.strong else n_arg  # tasks generated by this rank
    seed = (404 if hg19 else 42) + (0 if args.strong else rank)  # --strong: the same batch on every rank
    if hg19:
        (pool, q_off, qlen, t_off, tlen), w = synth_hg19_mixture_fast(n_batch, seed)
    else:
        (pool, q_off, qlen, t_off, tlen), w = synth_batch(n_batch, args.qlen, seed=seed), args.band
    cells_task = batch_cells(qlen, tlen, w)
    whole = None
    if args.strong and world > 1:  # LPT by cells; every rank computes the same partition
        whole = (pool, q_off, qlen, t_off, tlen, cells_task)
        shards = shard_tasks(cells_task, world)
        mine = shards[rank]
        q_off, qlen, t_off, tlen, cells_task = q_off[mine], qlen[mine], t_off[mine], tlen[mine], cells_task[mine]
    n = len(qlen)
    words, q_word, t_word = pack_batch(pool, q_off, qlen, t_off, tlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_word, t_word, qlen, tlen
    tasks["w"], tasks["zdrop"], tasks["flag"] = w, -1, 0
    cells_rank = int(cells_task.sum())

    # --inflight 2 (default): two device contexts, the steps alternate between them, each call on a host thread of its own
    # -- the planning of step i + 1 (0.6-1.6 ms before its first launch) and the tail of step i (last traceback, compaction)
    # overlap the other step's DP, as two lanes of the stage driver do (host/pipeline.cc).  Every step is still one whole
    # pass of the hot path over the batch, all K of them complete inside the timed region.
    engs = [sedef_amd.Extz2Engine(local, int(args.workspace_gib * (1 << 30))) for _ in range(args.inflight)]
    eng = engs[0]
    d_pool = torch.from_numpy(words.view(np.int32)).to(dev)
    cig_cap = int((qlen.astype(np.int64) + tlen + 2).sum())
    if not hg19:
        cig_cap = min(cig_cap, 256 * n)  # ~50 runs per task at 10 % divergence; overflow is an error
    # Two sets of output buffers in rotation: the result D2H of step i (and, at N > 1, the all-gatherv of its records and
    # CIGAR words: RCCL, on the communicator's stream) runs under the DP of step i+1, which writes the other set; every
    # copy and every gather is waited for inside the timed region.
    nsets = max(2, args.inflight)
    d_outs = [torch.empty(n * 16, dtype=torch.int32, device=dev) for _ in range(nsets)]
    d_cigs = [torch.empty(cig_cap, dtype=torch.int32, device=dev) for _ in range(nsets)]
    h_outs = [torch.empty(n * 16, dtype=torch.int32).pin_memory() for _ in range(nsets)]
    h_cigs = [torch.empty(cig_cap, dtype=torch.int32).pin_memory() for _ in range(nsets)]
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    # The engine works on a torch stream of its own, entered as the current stream: a gather's wait() then orders THIS
    # stream (and with it everything the engine launches: its internal streams fork from it) behind the collective that
    # last read the buffer set the step is about to overwrite.
    estreams = [torch.cuda.Stream(device=dev) for _ in engs]
    estream = estreams[0]
    cstream = torch.cuda.Stream(device=dev)  # result D2H
    copy_done = [None] * nsets

    gathers = []
    loopback_note = None
    if dist_on:
        gdev = torch.device("cpu") if debug_one_gpu else dev
        gathers = [ResultGatherV(gdev, torch.int32) for _ in range(nsets)]
        for g in gathers:  # (receive buffers at their high-water mark from the start: no step allocates)
            g.reserve(world * n * 16 + 16, world * cig_cap + 16)
            g.loopback = force_dist and os.environ.get("BENCH_LOOPBACK", "1") == "1"
    step_no = [0]
    last_used = [0] * nsets
    acc = {"dp": 0.0, "tb": 0.0, "cp": 0.0, "plan": 0.0, "call": 0.0, "launches": 0, "used": 0}
    pending = []  # steps started and not yet retired, oldest first: (buffer set, engine, future or result)
    pool_ex = None
    if len(engs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool_ex = ThreadPoolExecutor(len(engs))

    def compute(b, e):  # (a worker thread when two calls are in flight: the C call releases the GIL)
        return engs[e].align_batch_device(tasks, d_pool.data_ptr(), d_outs[b].data_ptr(), d_cigs[b].data_ptr(),
                                          cig_cap, want=want, stream=estreams[e].cuda_stream)

    def retire():  # main thread, in step order: the call has returned (its stream drained: the results are complete)
        b, e, fut = pending.pop(0)
        used = fut.result() if hasattr(fut, "result") else fut
        with torch.cuda.stream(estreams[e]):
            if gathers:  # all-gatherv of result records + CIGAR words, asynchronous
                if debug_one_gpu:
                    gathers[b].start(d_outs[b].cpu(), d_cigs[b][:used].cpu(), used)
                else:
                    gathers[b].start(d_outs[b], d_cigs[b], used)
        with torch.cuda.stream(cstream):  # result D2H of this rank's shard (pinned), asynchronous
            h_outs[b].copy_(d_outs[b], non_blocking=True)
            h_cigs[b][:used].copy_(d_cigs[b][:used], non_blocking=True)
            copy_done[b] = torch.cuda.Event()
            copy_done[b].record(cstream)
        last_used[b] = used
        acc["dp"] += engs[e].last_ms(0)
        acc["tb"] += engs[e].last_ms(1)
        acc["cp"] += engs[e].last_ms(2)
        acc["plan"] += engs[e].last_ms(4)
        acc["call"] += engs[e].last_ms(5)
        acc["launches"] += engs[e].last_launches()
        acc["used"] = used
        return used

    def step():
        i = step_no[0]
        step_no[0] += 1
        b, e = i % nsets, i % len(engs)
        while len(pending) >= len(engs):  # (the step that used this context and buffer set before)
            retire()
        with torch.cuda.stream(estreams[e]):
            if gathers:
                gathers[b].wait()  # the gather that last read this buffer set
            if copy_done[b] is not None:
                estreams[e].wait_event(copy_done[b])  # ... and the D2H that did
        pending.append((b, e, pool_ex.submit(compute, b, e) if pool_ex else compute(b, e)))
        if not pool_ex:
            retire()

    def sync():
        while pending:
            retire()
        for g in gathers:
            g.wait()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if gathers and gathers[0].loopback:
        try:  # (a send to oneself inside a group: if this build of the backend refuses it, the count all-gather still runs)
            step()
            sync()
        except Exception as ex:  # noqa: BLE001
            loopback_note = "refused by the backend: %s" % (str(ex).splitlines()[0][:160],)
            for g in gathers:
                g.loopback, g.handles = False, []
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    for k in acc:
        acc[k] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    dp_ms, tb_ms, cp_ms, launches, plan_ms, call_ms = acc["dp"], acc["tb"], acc["cp"], acc["launches"], acc["plan"], acc["call"]
    used = acc["used"]
    if dist_on:
        rdev = torch.device("cpu") if debug_one_gpu else dev
        tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([cells_rank], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot)
        cells_all = int(tot.item())
    else:
        cells_all = cells_rank

    lb = (step_no[0] - 1) % nsets
    res = h_outs[lb].numpy().view(sedef_amd.RESULT_DTYPE)
    assert int(res["n_cigar"].astype(np.int64).sum()) == used  # what came over PCIe is the whole result
    union_check = None
    if gathers:  # every rank holds every rank's results: this rank's own part of the last gather is what it computed
        pr, pc = gathers[lb].part(rank)
        assert torch.equal(pr.cpu(), h_outs[lb]) and torch.equal(pc.cpu(), h_cigs[lb][:used])
        if gathers[lb].loopback:  # what came back through the send / receive pair is what went out
            lr, lc = gathers[lb].loopback_part()
            assert torch.equal(lr.cpu(), h_outs[lb]) and torch.equal(lc.cpu(), h_cigs[lb][:used])
            loopback_note = "records and CIGAR words sent to this rank itself through the communicator: equal"
        if whole is not None and rank == 0:
            # --strong: the union of the shards against ONE GPU aligning the whole batch (untimed)
            wpool, wq_off, wqlen, wt_off, wtlen, _ = whole
            ww, wqw, wtw = pack_batch(wpool, wq_off, wqlen, wt_off, wtlen)
            wt = np.zeros(len(wqlen), sedef_amd.TASK_DTYPE)
            wt["q_off"], wt["t_off"], wt["qlen"], wt["tlen"], wt["w"], wt["zdrop"] = wqw, wtw, wqlen, wtlen, w, -1
            wd_pool = torch.from_numpy(ww.view(np.int32)).to(dev)
            wcap = int((wqlen.astype(np.int64) + wtlen + 2).sum())
            wd_out = torch.empty(len(wqlen) * 16, dtype=torch.int32, device=dev)
            wd_cig = torch.empty(wcap, dtype=torch.int32, device=dev)
            wused = eng.align_batch_device(wt, wd_pool.data_ptr(), wd_out.data_ptr(), wd_cig.data_ptr(), wcap, want=want)
            single = task_checksums(wd_out.cpu().numpy(), wd_cig[:wused].cpu().numpy())
            union = np.zeros(len(wqlen), np.uint64)
            for r in range(world):
                pr, pc = gathers[lb].part(r)
                union[shards[r]] = task_checksums(pr.cpu().numpy(), pc.cpu().numpy())
            assert np.array_equal(union, single), "union of the shards differs from the single-GPU run"
            union_check = "CIGAR+score checksum of all %d tasks: union of %d shards == single-GPU run" % (len(wqlen), world)
            del wd_pool, wd_out, wd_cig
    if rank == 0:
        alg = algorithmic_bytes(qlen, tlen, cells_task, res["n_cigar"])
        # Roofline of the dominant kernel.  The timed steps above run the batch as a pipeline of chunks whose DP
        # launches overlap each other and the traceback, so a launch's own duration is taken from one extra,
        # untimed pass of the same batch through a context created with pipeline = 0 (sdf_config): the whole batch in ONE
        # DP launch on one stream, HIP events around it on that stream (sdf_last_ms(0)).  The committed rocprofv3
        # summary (profiles/) is of `SDF_PIPELINE=0 python bench.py ...`, the same launch.
        iso_ms, iso_launches = dp_ms / max(args.steps, 1), max(launches // max(args.steps, 1), 1)
        roof_mode = "pipelined launches (union of the DP intervals)"
        headline = world == 1 and not hg19
        if headline:
            iso = sedef_amd.Extz2Engine(local, int(args.workspace_gib * (1 << 30)), config=dict(SDF_PIPELINE=0))
            for _ in range(2):  # warm-up + measured
                iso.align_batch_device(tasks, d_pool.data_ptr(), d_outs[0].data_ptr(), d_cigs[0].data_ptr(), cig_cap,
                                       want=want, stream=estream.cuda_stream)
            iso_ms, iso_launches = iso.last_ms(0), max(iso.last_launches(), 1)
            roof_mode = "one isolated DP launch of the whole batch (SDF_PIPELINE=0 pass, untimed)"
            del iso
        avg_launch_s = iso_ms / 1e3 / iso_launches
        bytes_per_launch = alg / iso_launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic = None
        valu = None
        traffic_note = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp) and headline and n == 100000 and w == 128 and args.qlen == 1000:
            # NOT measured in this run: read from the committed rocprofv3 PMC passes of this exact workload and launch
            # (profiles/README.md, profiles/collect.sh)
            pmc = json.load(open(tp))
            # ... and only while the kernel they were taken on is the kernel that ran: the JSON carries the sha256 of
            # extz2_pair.hip at collection time (profiles/make_traffic_json.py); after any edit of that file the
            # counters are stale and the line says null until profiles/collect.sh has run again
            stale = pmc.get("kernel_source_sha256") != source_sha256("sedef_amd/csrc/extz2_pair.hip")
            traffic_note = "profiles/hbm_traffic.json is older than sedef_amd/csrc/extz2_pair.hip: counters dropped" if stale else None
            if not stale:
                traffic = pmc.get("bytes_per_step") / iso_launches
            if not stale and pmc.get("valu_insts_per_step"):
                # What actually bounds the launch: VALU issue.  Counted wavefront VALU instructions over what the SIMDs can
                # issue in the launch's live-measured duration, against BOTH rates: one per 2 cycles (MI355X_MICROARCH.md:
                # wave64 on a SIMD-32) and one per 4.2 cycles (the measured rate of the VOP3P / VOP3 / DPP encodings
                # that make up the row, profiles/r01_ubench_issue_rates.txt; VOP2 issues at 2.5)
                props = torch.cuda.get_device_properties(local)
                simds = props.multi_processor_count * 4
                mhz = (getattr(props, "clock_rate", 0) or 2400000) / 1e3  # (2.4 GHz: MI355X peak engine clock)
                cycles = simds * mhz * 1e6 * avg_launch_s * iso_launches
                valu = {"insts_per_step": pmc["valu_insts_per_step"], "simds": simds, "clock_mhz": round(mhz, 1),
                        "frac_of_issue_peak_2cyc": round(pmc["valu_insts_per_step"] / (cycles / 2.0), 4),
                        "frac_of_issue_peak_4cyc": round(pmc["valu_insts_per_step"] / (cycles / 4.0), 4),
                        "frac_of_issue_peak_4p2cyc_measured_vop3p": round(pmc["valu_insts_per_step"] / (cycles / 4.2), 4),
                        "steady_row_by_encoding": encoding_split(),
                        "source": "committed profile: " + str(pmc.get("valu_source"))}
        value = cells_all * args.steps / dt / 1e9
        with torch.cuda.stream(estream):
            peak_measured = measured_copy_bandwidth(dev)
        # End to end over PCIe (never `value`): H2D of the packed sequences, the batch, D2H of the results, back to back
        # without overlap, untimed extra passes (include/sedef_hip.h: sdf_extz2_batch is this plus packing on the host)
        incl_pcie = None
        if world == 1 and not args.no_pcie_pass:
            h_pool = torch.from_numpy(words.view(np.int32)).pin_memory()
            d_pool2 = torch.empty_like(d_pool)
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                tp0 = time.perf_counter()
                with torch.cuda.stream(estream):
                    d_pool2.copy_(h_pool, non_blocking=True)
                    estream.synchronize()
                    u2 = eng.align_batch_device(tasks, d_pool2.data_ptr(), d_outs[0].data_ptr(), d_cigs[0].data_ptr(),
                                                cig_cap, want=want, stream=estream.cuda_stream)
                    h_outs[0].copy_(d_outs[0], non_blocking=True)
                    h_cigs[0][:u2].copy_(d_cigs[0][:u2], non_blocking=True)
                    estream.synchronize()
                times.append(time.perf_counter() - tp0)
            incl_pcie = round(cells_rank / min(times[1:]) / 1e9, 3)
        # One call at a time (what `--inflight 1` measures as `value`): the latency-bound figure next to the throughput one,
        # untimed extra -- every call returns when its stream has drained, its results go out over pinned D2H like in the
        # timed steps, all waited for inside the clock
        one_call = None
        if world == 1 and len(engs) > 1:
            k1 = max(3, min(args.steps, 10))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k1):
                u1 = engs[0].align_batch_device(tasks, d_pool.data_ptr(), d_outs[0].data_ptr(), d_cigs[0].data_ptr(), cig_cap,
                                                want=want, stream=estreams[0].cuda_stream)
                with torch.cuda.stream(cstream):
                    h_outs[0].copy_(d_outs[0], non_blocking=True)
                    h_cigs[0][:u1].copy_(d_cigs[0][:u1], non_blocking=True)
            torch.cuda.synchronize()
            one_call = round(cells_rank * k1 / (time.perf_counter() - t1) / 1e9, 3)
        if hg19:
            wl = ("configs[3]: hg19-shaped task mixture, %d DP tasks %s, all w=-1 (59 %% <=100 cells, 40 %% <=1e4, 1.2 %% "
                  "500x500, 0.06 %% <=1000^2, 0.06 %% up to 6000^2), affine gap 5/-4/40/1, CIGAR+score+counts"
                  % (n_arg, "in one batch sharded by cells" if args.strong else "per GPU"))
        else:
            wl = ("configs[1]: %d synthetic %dx~%d DP tasks %s, band=%d, affine gap 5/-4/40/1, CIGAR+score+counts"
                  % (n_arg, args.qlen, args.qlen, "in one batch sharded by cells" if args.strong else "per GPU", w))
        line = {
            "metric": "aligned DP cells/sec (Gcell/s) on `sedef align` batch",
            "value": round(value, 3), "unit": "Gcell/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "i8",
            "data": "synthetic",
            "value_incl_pcie": incl_pcie,
            "value_one_call_in_flight": one_call,
            "timed_region": "planning + DP + traceback + CIGAR compaction + result D2H (pinned, double-buffered under the "
                            "next step)" + (" + all-gatherv of records and CIGARs" if dist_on else "") +
                            ("; two batch calls in flight on two device contexts: a step's planning and tail run under the "
                             "other step's DP" if len(engs) > 1 else ""),
            "config": {"workload": wl, "tasks_this_rank": n, "band": int(w) if np.ndim(w) == 0 else "mixed",
                       "cells_per_step_this_rank": cells_rank, "cells_per_step_all_ranks": cells_all,
                       "parallelism": "task-sharded x%d + all-gatherv of result records" % world,
                       "calls_in_flight": len(engs),
                       "union_check": union_check,
                       "dist_backend": (dist.get_backend() if dist_on else None),
                       "forced_dist_one_rank": force_dist or None, "loopback": loopback_note},
            "ms_per_call": round(call_ms / args.steps, 3),
            "kernel_ms_per_step": {"what": "sums over the step's batch CALL of the intervals its own events and clocks measured "
                                           "(DP = union of that call's DP launches); with calls_in_flight > 1 the calls "
                                           "overlap, so these per-call figures can exceed ms_per_step -- ms_per_call is "
                                           "the wall time of one call, ms_per_step the spacing of completed steps",
                                   "dp_launches": launches / args.steps,
                                   "dp": round(dp_ms / args.steps, 3), "traceback": round(tb_ms / args.steps, 3),
                                   "compact": round(cp_ms / args.steps, 3),
                                   "host_planning": round(plan_ms / args.steps, 3),
                                   "host_call_total": round(call_ms / args.steps, 3)},
            # `bound` / `achieved` / `peak` / `frac` are the HBM roofline SURVEY 8(d) defines for this path (algorithmic bytes
            # over the launch's duration).  It is NOT what limits the launch: `binding_limit` says what does -- the vector
            # ALUs' issue slots, from the launch's counted VALU instructions at the measured issue cost of their encodings.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "binding_limit": ({"bound": "valu_issue", "frac": valu["frac_of_issue_peak_4cyc"],
                                            "how": "counted wavefront VALU instructions of the launch x 4 cycles (a wave64 vector "
                                                   "instruction holds its SIMD's ALU for four cycles: the rate every DP launch "
                                                   "of this library that fills the device was measured at, profiles/"
                                                   "r06_chain_pmc.txt; a lone stream of packed / VOP3 / DPP encodings was "
                                                   "measured at 4.2, valu_issue.frac_of_issue_peak_4p2cyc_measured_vop3p) over "
                                                   "SIMDs x nominal clock x the launch's live-measured duration; 1.0 = every "
                                                   "issue slot of every SIMD taken",
                                            "hbm_frac_is": "%.2f of the HBM peak because the kernel is bound elsewhere: its HBM "
                                                           "traffic (`traffic`) is %.2f x the algorithmic bytes, nothing is re-read "
                                                           "(what exceeds 1.0 is register spills at the starts of 16-row blocks)"
                                                           % (achieved / HBM_PEAK_GBS, (traffic or 0) / max(bytes_per_launch, 1))}
                                           if valu else {"bound": "valu_issue", "frac": None,
                                                         "how": "no committed counters for this workload / kernel source (see "
                                                                "traffic_source); the DP kernels of this path are VALU-issue "
                                                                "bound on every profiled shape (DESIGN.md 5)"}),
                         "peak_measured": round(peak_measured, 1), "frac_of_measured": round(achieved / peak_measured, 5),
                         "peak_measured_how": "device-to-device copy of 1 GiB in this run, read + write bytes over the best of 5",
                         "traffic": traffic,
                         "traffic_source": ("committed profile (profiles/hbm_traffic.json), not measured in this run" if traffic
                                            else traffic_note),
                         "kernel": "extz2_pair_kernel<3> (extz2 DP)" if not hg19 else "all DP launches of the mixture",
                         "measured_on": roof_mode,
                         "launches": iso_launches, "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch), "valu_issue": valu},
        }
        if not args.no_cpu_baseline:
            # (at every N: rank 0's shard on this box's host cores while the other ranks wait at the barrier below -- the
            # ratio to quote is value / cpu_baseline.value per cpu_baseline.cores cores)
            cb = cpu_baseline(pool, q_off, qlen, t_off, tlen, cells_task, w)
            line["cpu_baseline"] = cb
            # the north star's ">= 50 x per node" needs a stated denominator: the measured one is `cores` host cores (a
            # cgroup quota on this pool); the whole-host figure is a LINEAR projection from the per-core rate
            per_gpu = value / world
            line["vs_cpu"] = {
                "x_per_cpu_cores_measured": round(value / cb["value"], 2), "measured_on_cores": cb["cores"],
                "whole_host_linear_projection_gcells": round(cb["per_core"] * cb["host_cores_total"], 3),
                "x_vs_whole_host_linear_projection": round(value / (cb["per_core"] * cb["host_cores_total"]), 3),
                "x_vs_whole_host_at_8_gpus_if_linear": round(8 * per_gpu / (cb["per_core"] * cb["host_cores_total"]), 2),
                "note": "value / cpu_baseline.value is per `measured_on_cores` cores; the projections assume the reference "
                        "kernel scales linearly to host_cores_total logical CPUs and the GPUs to 8 (neither measured here)"}
        if world == 1 and os.environ.get("BENCH_NO_SPOT_CHECK") != "1":
            line["spot_check"] = spot_check(pool, q_off, qlen, t_off, tlen, w, res, h_cigs[lb].numpy().view(np.uint32))
        if headline and not args.no_stage and os.environ.get("BENCH_NO_STAGE") != "1":
            line["stage"] = stage_block()
        print(json.dumps(line))
    if dist_on:
        dist.barrier(group=quiet) if quiet is not None else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
