"""The batch planner (sedef_amd/csrc/sdf_plan.hip) without a GPU: cut_batch + plan_chunk through the library's
sdf_debug_plan hook.  Invariants of a plan: every runnable task is planned exactly once, direction-flag regions and CIGAR
staging slots are disjoint and inside the chunk's workspace slice, paired tasks share their geometry, heavy tasks leave
the chunk rotation, and planning on several threads gives the same plan as planning on one."""
import ctypes as C

import numpy as np
import pytest


def _plan(tasks, want=3, ws=64 << 30, lds=160 * 1024, threads=0, mat=None, gapo=40, gape=1):
    import sedef_amd
    from sedef_amd import extz2
    lib = sedef_amd.load_library()
    sc = extz2._scoring(extz2.sedef_mat() if mat is None else mat, gapo, gape)
    n = len(tasks)
    per_task = np.zeros((n, 7), np.int64)
    per_chunk = np.zeros((64, 5), np.int64)
    nch = C.c_size_t(0)
    lib.sdf_debug_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = lib.sdf_debug_plan(C.byref(sc), tasks.ctypes.data, n, want, ws, lds, threads, per_task.ctypes.data,
                            per_chunk.ctypes.data, 64, C.byref(nch))
    return rc, per_task, per_chunk[:nch.value]


def _tasks(qlen, tlen, w=-1, flag=0, zdrop=-1):
    from sedef_amd.extz2 import TASK_DTYPE
    t = np.zeros(len(qlen), TASK_DTYPE)
    t["qlen"], t["tlen"], t["w"], t["flag"], t["zdrop"] = qlen, tlen, w, flag, zdrop
    return t


def _dir_need(q, t, bs, nreg, pad):
    nblk = (q + t - 1 + 15) // 16
    if pad == 2:
        return nblk * nreg * 512
    if pad in (5, 7, 9, 10):
        return None  # stripe / strip layouts: checked by disjointness only (strips: test_strip_pairs_fit_...)
    if nreg:
        return nblk * nreg * 1024
    return None


def _check(tasks, per_task, per_chunk):
    runs = (tasks["qlen"] > 0) & (tasks["tlen"] > 0)
    assert ((per_task[:, 0] >= 0) == runs).all()
    for ci in range(len(per_chunk)):
        sel = np.flatnonzero(per_task[:, 0] == ci)
        assert len(sel) == per_chunk[ci, 1]
        assert per_chunk[ci, 3] <= per_chunk[ci, 4]
        offs = np.sort(per_task[sel, 4])
        assert len(np.unique(offs)) == len(offs) or (tasks["flag"][sel] & 1).any()
        assert offs.max(initial=0) < max(per_chunk[ci, 3], 1)
        for k in sel[:2000]:
            need = _dir_need(int(tasks["qlen"][k]), int(tasks["tlen"][k]), *[int(x) for x in per_task[k, 1:4]])
            if need is not None:
                nxt = offs[np.searchsorted(offs, per_task[k, 4], side="right"):]
                assert len(nxt) == 0 or per_task[k, 4] + need <= nxt[0]
    slots = per_task[runs, 5]
    assert len(np.unique(slots)) == len(slots)  # CIGAR staging slots: one per task
    order = np.argsort(slots)
    cap = (tasks["qlen"][runs].astype(np.int64) + tasks["tlen"][runs] + 2)[order]
    assert (slots[order][1:] >= slots[order][:-1] + cap[:-1]).all()
    part = per_task[:, 6]
    for k in np.flatnonzero(part >= 0):
        p = int(part[k])
        assert part[p] == k and per_task[p, 0] == per_task[k, 0] and per_task[p, 1] == per_task[k, 1]
        if per_task[k, 1] in (500, 604, 608):  # strip kernels: two tasks of any geometry with as many column blocks
            bw = 64 * (8 if per_task[k, 1] == 500 else int(per_task[k, 2]))
            assert (tasks["tlen"][p] + bw - 1) // bw == (tasks["tlen"][k] + bw - 1) // bw
            lo, hi = sorted((int(tasks["qlen"][p]), int(tasks["qlen"][k])))
            assert hi <= lo + lo // 8 + 64  # (the flag bound of cut_batch counts on it)
        elif 130 <= per_task[k, 1] < 140:  # mixed pair (extz2_pair.hip, MIXED): one band and flag set, any lengths
            assert tasks["w"][p] == tasks["w"][k] and tasks["flag"][p] == tasks["flag"][k]
            assert per_task[k, 2] == per_task[p, 2]  # ... in the same number of window registers
        else:
            assert (tasks["qlen"][p], tasks["tlen"][p], tasks["w"][p]) == (tasks["qlen"][k], tasks["tlen"][k], tasks["w"][k])


def test_headline_batch_is_chunked_paired_and_deterministic():
    rng = np.random.default_rng(1)
    n = 100000
    t = _tasks(np.full(n, 1000), 1000 + rng.integers(-25, 25, n), w=128)
    rc, pt, pc = _plan(t)
    assert rc == 0 and len(pc) >= 4 and not pc[:, 0].any()
    assert (pt[:, 1] == 103).all() and (pt[:, 6] >= 0).all()  # extz2_pair_kernel<3>, every task has a partner
    assert pc[0, 1] <= 8192 < pc[1, 1]  # a small first chunk: the GPU starts early
    _check(t, pt, pc)
    rc2, pt2, pc2 = _plan(t, threads=4)
    assert rc2 == 0 and (pt2 == pt).all() and (pc2 == pc).all()


def test_mixture_with_heavy_tasks_and_empty_tasks():
    rng = np.random.default_rng(2)
    n = 80000
    q = rng.integers(1, 220, n)
    tl = np.maximum(1, q + rng.integers(-8, 8, n))
    big = rng.choice(n, 60, replace=False)
    q[big] = rng.integers(1500, 9000, 60)
    tl[big] = q[big] + rng.integers(-100, 100, 60)
    q[rng.choice(n, 50, replace=False)] = 0  # reference early return: no work, reset result
    t = _tasks(q, tl)
    rc, pt, pc = _plan(t, threads=3)
    assert rc == 0
    heavy = np.flatnonzero(pc[:, 0])
    assert len(heavy) >= 1 and (heavy == np.arange(len(heavy))).all()  # heavy chunks are launched first
    hv_tasks = np.isin(pt[:, 0], heavy)
    assert hv_tasks.sum() >= 40 and (t["qlen"][hv_tasks] >= 700).all()
    assert set(np.unique(pt[hv_tasks, 1])) <= {301, 302, 304, 2256, 3024, 2001, 104, 114, 106, 116, 108, 118, 8, 18}
    _check(t, pt, pc)
    rc1, pt1, pc1 = _plan(t, threads=0)
    assert (pt1 == pt).all() and (pc1 == pc).all()


def test_kernel_choice_by_request():
    t = _tasks([500, 500, 500, 500, 3000, 70000], [500, 500, 500, 500, 3000, 300], w=[-1, -1, 64, 64, -1, -1],
               zdrop=[-1, 100, -1, -1, -1, -1], flag=[0, 0, 0, 2, 0, 0])
    rc, pt, pc = _plan(t)
    assert rc == 0
    assert pt[0, 1] in (301, 302, 304) and pt[1, 1] in (64, 256, 1024)  # full band from 400 target bases up: stripe kernel;
    #                                                                    z-drop needs every ksw_extz_t field: general kernel
    assert pt[2, 1] in (102, 103) and pt[3, 1] in (64, 256)        # right-aligned gaps: general kernel
    assert pt[4, 1] in (301, 302, 304) and pt[4, 1] == pt[0, 1]   # wide full-band: stripe kernel, one width per chunk
    rc, pt, pc = _plan(_tasks([300, 390], [300, 390]))
    assert rc == 0 and (pt[:, 1] // 10 == 10).all()               # short full-band tasks: two to a wavefront (pair kernel)
    rc, pt, pc = _plan(t, want=7)                                 # every field wanted: no register-resident kernel
    assert rc == 0 and (pt[:, 1] >= 64).all() and not np.isin(pt[:, 1], (101, 102, 103, 104, 106, 108, 201, 202, 204, 301, 302, 304)).any()
    rc, pt, _ = _plan(_tasks([300, 300], [300, 300], flag=[4, 8]))
    assert rc == 0 and (pt[:, 1] == 256).all()  # generic scoring / approximate max: the general kernel
    rc, _, _ = _plan(_tasks([10], [10], flag=[0x100]))
    assert rc == -3  # SDF_ERR_UNSUPPORTED: not a flag of this kernel
    rc, pt, _ = _plan(_tasks([300, 0], [300, 5]), mat=np.array([1] + [-100] * 24, np.int8), gapo=1, gape=1)
    assert rc == 0 and (pt[:, 0] == -1).all()  # degenerate scoring: the reference returns before any work
    # the window kernels' 32-bit differences want every fresh score byte in q .. 127 (sdf_api.hip: scoring_gates)
    from sedef_amd import extz2
    t2 = _tasks([300, 300, 1000, 1000], [300, 300, 1000, 1000], w=[-1, -1, 128, 128])
    for (ma, mi, go, ge), fast in (((5, -10, 10, 0), True), ((5, -11, 10, 0), False), ((5, -4, 60, 1), True),
                                   ((5, -4, 61, 1), False), ((5, -4, 40, 1), True)):
        rc, pt, _ = _plan(t2, mat=extz2.sedef_mat(ma, mi), gapo=go, gape=ge)
        assert rc == 0
        on_general = np.isin(pt[:, 1], (64, 256, 1024))
        assert on_general.all() != fast and (not fast or not on_general.any()), (ma, mi, go, ge, pt[:, 1])


def test_stripe_kernel_routing_and_widths(monkeypatch):
    """Full-band tasks from 400 target bases up and long banded tasks take the stripe kernels; the stripe width of a chunk
    follows its work (one long task: 128 positions, thousands of tasks: 512), the banded kernel's the target length."""
    rc, pt, _ = _plan(_tasks([6000], [6000]))
    assert rc == 0 and pt[0, 1] == 301                        # one long chain: the narrowest stripes
    rc, pt, _ = _plan(_tasks([1000] * 6000, [1000] * 6000))
    assert rc == 0 and (pt[:, 1] == 608).all()                # many tasks: chained strips, eight columns a lane (6,000 wavefronts)
    rc, pt, _ = _plan(_tasks([1000] * 4000, [1000] * 4000))
    assert rc == 0 and (pt[:, 1] == 604).all()                # fewer: four columns a lane
    monkeypatch.setenv("SDF_NO_STRIP", "1")
    rc, pt, _ = _plan(_tasks([1000] * 4000, [1000] * 4000))
    assert rc == 0 and (pt[:, 1] == 304).all()                # ... without the strip kernels: the widest stripes
    t = _tasks([7000, 7000, 3000, 20000, 33000, 1000, 2500, 12000], [7000, 7100, 3100, 20000, 33000, 1000, 2700, 11000],
               w=[64, 128, 16, 200, 200, 128, 100, 300])
    rc, pt, pc = _plan(t)
    assert rc == 0
    assert pt[3, 1] == 401                                    # windows of more than 192 slots, 4000+ anti-diagonals
    assert pt[4, 1] == 402                                    # more than 254 stripes of 128 positions: 256
    assert pt[0, 1] // 10 in (10, 11) and pt[1, 1] // 10 in (10, 11)  # windows of up to 192 slots that reach the corner: pair kernel
    assert pt[2, 1] == 401 and pt[7, 1] == 401                # bands that run out: banded stripes whatever their width
    assert pt[5, 1] // 10 in (10, 11) and pt[6, 1] == 401     # too short / just long enough (and its band runs out)
    _check(t, pt, pc)


def test_two_pass_cut_with_early_heavy_chunks(monkeypatch):
    """Batches of 400,000 tasks and more are cut in two passes -- the big tasks first, whose heavy chunks are planned (and,
    on a device, launched) from a callback while the rest of the batch is read: every runnable task is still planned exactly
    once, the heavy chunks come first, the buffers sized by the first pass's upper bounds hold, and the plan is a valid one
    (regions and staging slots disjoint) -- the same invariants as the one-pass cut, which marks the same tasks heavy."""
    rng = np.random.default_rng(5)
    n = 420000
    q = rng.integers(1, 210, n)
    tl = np.maximum(1, q + rng.integers(-6, 6, n))
    big = rng.choice(n, 700, replace=False)
    q[big[:300]] = rng.integers(1200, 6000, 300)
    tl[big[:300]] = q[big[:300]] + rng.integers(-60, 60, 300)
    q[big[300:]] = 500
    tl[big[300:]] = 500 + rng.integers(-10, 10, 400)
    q[rng.choice(n, 40, replace=False)] = 0
    t = _tasks(q, tl)
    rc0, pt0, pc0 = _plan(t, threads=3)
    assert rc0 == 0
    monkeypatch.setenv("SDF_DEBUG_PLAN_EARLY", "1")
    rc, pt, pc = _plan(t, threads=3)
    assert rc == 0
    heavy = np.flatnonzero(pc[:, 0])
    assert len(heavy) >= 1 and (heavy == np.arange(len(heavy))).all()
    _check(t, pt, pc)
    in_heavy, in_heavy0 = np.isin(pt[:, 0], heavy), np.isin(pt0[:, 0], np.flatnonzero(pc0[:, 0]))
    assert in_heavy.sum() >= 650 and (in_heavy == in_heavy0).all()
    assert ((pt[:, 0] >= 0) == (pt0[:, 0] >= 0)).all()  # the same tasks run


def _mm8_like(rng, n):
    q = np.exp(rng.uniform(np.log(200), np.log(20000), n)).astype(np.int64)
    t = np.maximum(1, q + (rng.normal(0, 1, n) * np.sqrt(0.04 * q)).astype(np.int64))
    cut = (np.arange(n) % 20 == 0) & (t > 6000)
    t = np.where(cut, t - rng.integers(1000, 5001, n), t)
    return _tasks(q, t, w=rng.choice([64, 128, 256, 512], n))


def test_banded_tasks_of_all_lengths_mixed_pairs_and_fewer_chunks():
    """BASELINE configs[4]'s shape (lengths log-uniform 200..20,000, bands 64..512): in a batch large enough to fill the
    device the banded tasks whose band reaches the end run two to a wavefront whatever their lengths (launch classes 132..139:
    the MIXED pair kernel with 2..9 window registers, w = 512 included), the long ones among them leave the banded stripe
    kernel, tasks whose band runs out keep it; and the batch is cut into FEWER chunks than its size alone would give -- a
    launch lasts as long as its longest chain of rows, so a chunk must hold several chains' worth of work.  A batch of
    3,000 such tasks stays as it was: long tasks on banded stripes, no mixed pairs."""
    rng = np.random.default_rng(11)
    t = _mm8_like(rng, 40000)
    rc, pt, pc = _plan(t)
    assert rc == 0 and 1 <= len(pc) <= 3 and not pc[:, 0].any()  # (by size alone: a small first chunk + four)
    assert pc[:, 1].min() > 0.4 * pc[:, 1].max()                  # ... of about equal size (whole blocks of 4,096 tasks), no small first chunk
    mixed = (pt[:, 1] >= 130) & (pt[:, 1] < 140)
    assert mixed.sum() > 34000 and (pt[mixed, 6] >= 0).all()
    assert set(np.unique(pt[mixed, 1])) >= {132, 133, 135, 139}
    w512 = (t["w"] == 512) & mixed
    assert (pt[w512, 1] == 139).all() and w512.sum() > 7000
    runs_out = np.abs(t["qlen"].astype(np.int64) - t["tlen"]) > t["w"]
    long_out = runs_out & (t["qlen"] + t["tlen"] >= 4001)
    assert long_out.sum() > 200 and (pt[long_out, 1] // 100 == 4).all() and not mixed[runs_out].any()
    _check(t, pt, pc)
    rc2, pt2, pc2 = _plan(t, threads=3)
    assert rc2 == 0 and (pt2 == pt).all() and (pc2 == pc).all()
    small = _mm8_like(rng, 3000)
    rc, pt, pc = _plan(small)
    assert rc == 0 and len(pc) == 1 and not ((pt[:, 1] >= 130) & (pt[:, 1] < 140)).any()
    long_wide = (small["qlen"] + small["tlen"] >= 4001) & (small["w"] >= 256)
    assert (pt[long_wide, 1] // 100 == 4).all()
    _check(small, pt, pc)




@pytest.mark.parametrize("cols", ["0", "4", "8"])
def test_strip_pairs_fit_the_regions_their_bounds_reserved(monkeypatch, cols):
    """Strip and chained-strip tasks share a flag region per PAIR whose size the cut does not know when it reserves each task's
    share (cut_batch: 0.54 rows + 60 records per block and task, pairs of at most 1.125 m + 64 rows): plan_chunk refuses a
    chunk whose placed flags overrun the region ("direction-flag region overflow").  Full-band tasks of every size the two
    kernels take, sorted and unsorted, on workspaces that hold the batch in one chunk and in several."""
    monkeypatch.setenv("SDF_STRIP_COLS", cols)
    monkeypatch.setenv("SDF_STRIP_ALWAYS", "1")  # (a chunk of few tasks would take the stripe kernels)
    rng = np.random.default_rng(66 + int(cols))
    for trial in range(6):
        n = 6000
        tl = np.exp(rng.uniform(np.log(257), np.log(7000), n)).astype(np.int64)
        if trial % 3 == 0:   # rows far from the columns: pairs of very different heights among the neighbours
            ql = np.exp(rng.uniform(np.log(64), np.log(7000), n)).astype(np.int64)
        elif trial % 3 == 1:
            ql = np.maximum(64, tl + rng.integers(-200, 200, n))
        else:                # few distinct heights, each just beyond the pairing rule of the one before
            ql = np.array([64, 137, 219, 311, 414, 530, 661, 808, 973, 1159])[rng.integers(0, 10, n)]
        t = _tasks(ql, tl)
        for ws in (64 << 30, 2 << 30, 600 << 20):
            rc, pt, pc = _plan(t, ws=ws)
            assert rc == 0, (trial, ws)
            assert ((pt[:, 0] >= 0)).all()
            assert (pc[:, 3] <= pc[:, 4]).all()     # flags placed <= the region the bounds reserved
            assert np.isin(pt[:, 1], (500, 604, 608)).sum() > n // 2
            _check(t, pt, pc)
            # every strip pair's region ends before the next region of its chunk begins
            for ci in range(len(pc)):
                sel = np.flatnonzero((pt[:, 0] == ci))
                offs = np.sort(pt[sel, 4])
                for k in sel[np.isin(pt[sel, 1], (500, 604, 608))][:1500]:
                    p = int(pt[k, 6])
                    if pt[k, 4] > pt[p, 4]:
                        continue  # (the pair's region starts at the other one's offset)
                    c8 = 8 if pt[k, 1] == 500 else int(pt[k, 2])
                    qm, tm = max(int(ql[k]), int(ql[p])), max(int(tl[k]), int(tl[p]))
                    blocks = (tm + 64 * c8 - 1) // (64 * c8)
                    nrec = (qm + 64) >> 1 if c8 == 4 else qm + 63
                    need = blocks * nrec * (256 if p == k else 512)
                    if pt[k, 1] != 500:
                        need = ((need + 255) & ~255) + (((blocks - 1) * (qm + 64) * 4 + 64 + 255) & ~255)
                    nxt = offs[np.searchsorted(offs, pt[k, 4] + 4, side="right"):]
                    assert (len(nxt) == 0 or pt[k, 4] + need <= nxt[0]) and pt[k, 4] + need <= pc[ci, 3]
