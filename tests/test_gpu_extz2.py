"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the CPU oracle."""
import numpy as np
import pytest

from oracle.binding import mutate, random_codes
from util import FIELDS, check_case, cigar_to_str, codes, sedef_mat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import sedef_amd
    return sedef_amd.Extz2Engine(0)


@pytest.fixture(scope="module")
def solo_engine():
    """Context with SDF_NO_PAIR=1: one task per wavefront (extz2_wave.hip) instead of the pair kernel."""
    import sedef_amd
    return sedef_amd.Extz2Engine(0, config=dict(SDF_NO_PAIR=1))


def _engine_with_env(**env):
    """A context whose settings differ from the environment's by `env` (SDF_* names): through the C ABI's configuration
    struct (sdf_create_cfg), never through os.environ."""
    import sedef_amd
    return sedef_amd.Extz2Engine(0, config=env)


@pytest.fixture(scope="module")
def onetask_engine():
    """Context with the banded stripe kernel off: long banded tasks stay on the one-task kernels (pair / wave / general)."""
    return _engine_with_env(SDF_BSTRIPE_MIN_ROWS=0)


@pytest.fixture(scope="module")
def onetask_solo_engine():
    return _engine_with_env(SDF_BSTRIPE_MIN_ROWS=0, SDF_NO_PAIR=1)


def _rec_to_dict(r, cig):
    d = {k: int(r[k]) for k in FIELDS}
    d["cigar"] = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
    return d


GPU_FLAGS_OK = lambda f: True  # noqa: E731  (generic scoring and the approximate modes run on the general kernel)


def test_golden_vectors_single_calls(engine, golden_cases):
    import sedef_amd
    n = 0
    for c in golden_cases:
        if not GPU_FLAGS_OK(c["flag"]):
            continue
        got = sedef_amd.ksw_extz2(codes(c["q"]), codes(c["t"]), 5, sedef_mat(c["match"], c["mismatch"]),
                                  c["gapo"], c["gape"], c["w"], c["zdrop"], c["flag"], engine=engine)
        check_case(got, c)
        n += 1
    assert n == len(golden_cases) >= 284


def test_golden_vectors_one_batch(engine, golden_cases):
    sel = [c for c in golden_cases if GPU_FLAGS_OK(c["flag"]) and (c["match"], c["mismatch"], c["gapo"], c["gape"]) == (5, -4, 40, 1)]
    pairs = [(codes(c["q"]), codes(c["t"])) for c in sel]
    res, cig = engine.align_pairs(pairs, w=[c["w"] for c in sel], zdrop=[c["zdrop"] for c in sel],
                                  flag=[c["flag"] for c in sel])
    for c, r in zip(sel, res):
        check_case(_rec_to_dict(r, cig), c)


def test_unsupported_flags_are_rejected(engine):
    import sedef_amd
    q = np.zeros(10, np.uint8)
    for f in (0x100, 0x200):  # not KSW_EZ_* bits of this kernel
        with pytest.raises(sedef_amd.SdfError):
            engine.align_pairs([(q, q)], flag=f)


def test_fuzz_vs_oracle(engine, oracle):
    rng = np.random.default_rng(2024)
    pairs, kws = [], []
    for _ in range(1500):
        q = random_codes(rng, int(rng.integers(1, 600)), 0.02 if rng.random() < 0.3 else 0.0)
        d = rng.random() * 0.15
        t = mutate(rng, q, d, d / 3, d / 3)
        if rng.random() < 0.3:
            k, L = int(rng.integers(0, len(t))), int(rng.integers(1, 90))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]])
        if rng.random() < 0.1:
            t = random_codes(rng, int(rng.integers(1, 600)))
        pairs.append((q, t))
        kws.append(dict(w=int(rng.choice([-1, -1, 0, 1, 2, 7, 15, 16, 17, 31, 32, 33, 64, 128])),
                        flag=int(rng.choice([0, 0, 0, 0x02, 0x40, 0x80, 0x01, 0xC2])),
                        zdrop=int(rng.choice([-1, -1, 30, 200]))))
    res, cig = engine.align_pairs(pairs, w=[k["w"] for k in kws], zdrop=[k["zdrop"] for k in kws],
                                  flag=[k["flag"] for k in kws])
    for (q, t), kw, r in zip(pairs, kws, res):
        exp = oracle.extz2(q, t, **kw)
        got = _rec_to_dict(r, cig)
        for f in FIELDS:
            assert got[f] == exp[f], (f, kw, len(q), len(t))
        assert cigar_to_str(got["cigar"]) == cigar_to_str(exp["cigar"]), (kw, len(q), len(t))
        full_path = not (kw["flag"] & (0x80 | 0x40 | 0x01)) and not exp["zdropped"]
        if full_path:  # column counters (populate_nice_alignment restated) over a whole forward CIGAR
            assert {k: int(r[k]) for k in ("matches", "mismatches", "gaps", "gap_bases")} == \
                oracle.counts(exp["cigar"], q, t)


def test_fuzz_other_scorings(engine, oracle):
    rng = np.random.default_rng(77)
    for _ in range(60):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 70)), int(rng.integers(0, 6))
        pairs = []
        for _ in range(12):
            q = random_codes(rng, int(rng.integers(1, 300)), 0.03)
            d = rng.random() * 0.2
            pairs.append((q, mutate(rng, q, d, d / 3, d / 3)))
        w = int(rng.choice([-1, 3, 16, 40, 100]))
        res, cig = engine.align_pairs(pairs, w=w, mat=sedef_mat(ma, mi), gapo=go, gape=ge)
        for (q, t), r in zip(pairs, res):
            exp = oracle.extz2(q, t, mat=sedef_mat(ma, mi), gapo=go, gape=ge, w=w)
            got = _rec_to_dict(r, cig)
            for f in FIELDS:
                assert got[f] == exp[f], (f, ma, mi, go, ge, w)
            assert cigar_to_str(got["cigar"]) == cigar_to_str(exp["cigar"])


def test_sedef_shapes_full_band(engine, oracle):
    """SEDEF's real call shapes: w=-1, gap fills <=210x209, 500x500 sides, <=1000^2 (src/align.cc:129,159,235)."""
    rng = np.random.default_rng(5)
    pairs = []
    for ql, tl in [(210, 209), (37, 3), (3, 180), (500, 500), (500, 431), (1000, 1000), (1000, 640),
                   (977, 1000), (1, 1), (1, 300), (2000, 1800), (16, 16), (17, 15)]:
        q = random_codes(rng, ql)
        t = mutate(rng, q)
        t = t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])
        pairs.append((q, t))
    res, cig = engine.align_pairs(pairs)
    for (q, t), r in zip(pairs, res):
        exp = oracle.extz2(q, t)
        got = _rec_to_dict(r, cig)
        for f in FIELDS:
            assert got[f] == exp[f], (f, len(q), len(t))
        assert cigar_to_str(got["cigar"]) == cigar_to_str(exp["cigar"])
        assert {k: int(r[k]) for k in ("matches", "mismatches", "gaps", "gap_bases")} == \
            oracle.counts(exp["cigar"], q, t)


# ---- register-resident wave kernel (taken when only CIGAR / score / mte are requested) ----
FAST_FIELDS = ("score", "mte", "mte_q", "zdropped")


def _check_fast(engine, oracle, pairs, ws, flags=None, **sc):
    import sedef_amd
    flags = flags or [0] * len(pairs)
    res, cig = engine.align_pairs(pairs, w=ws, flag=flags, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE,
                                  **sc)
    for (q, t), w, f, r in zip(pairs, ws, flags, res):
        exp = oracle.extz2(q, t, w=w, flag=f, **sc)
        got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        for fld in FAST_FIELDS:
            assert int(r[fld]) == exp[fld], (fld, w, len(q), len(t), int(r[fld]), exp[fld])
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"]), (w, len(q), len(t))
        if not exp["zdropped"] and not (f & 0x81):
            assert {k: int(r[k]) for k in ("matches", "mismatches", "gaps", "gap_bases")} == \
                oracle.counts(exp["cigar"], q, t)


def test_wave_kernel_golden(solo_engine, golden_cases):
    import sedef_amd
    sel = [c for c in golden_cases if (c["flag"] & ~0x81) == 0 and c["zdrop"] < 0
           and (c["match"], c["mismatch"], c["gapo"], c["gape"]) == (5, -4, 40, 1)]
    assert len(sel) > 150
    pairs = [(codes(c["q"]), codes(c["t"])) for c in sel]
    res, cig = solo_engine.align_pairs(pairs, w=[c["w"] for c in sel], flag=[c["flag"] for c in sel],
                                  want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    for c, r in zip(sel, res):
        exp = c["expect"]
        got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        for fld in FAST_FIELDS:
            assert int(r[fld]) == exp[fld], (c["tag"], fld, c["w"], len(c["q"]), len(c["t"]))
        assert cigar_to_str(got) == exp["cigar"], (c["tag"], c["w"], len(c["q"]), len(c["t"]))


def test_wave_kernel_fuzz_banded(solo_engine, oracle):
    rng = np.random.default_rng(99)
    pairs, ws = [], []
    for _ in range(1200):
        q = random_codes(rng, int(rng.integers(1, 900)), 0.02 if rng.random() < 0.3 else 0.0)
        d = rng.random() * 0.15
        t = mutate(rng, q, d, d / 3, d / 3)
        if rng.random() < 0.4:
            k, L = int(rng.integers(0, len(t))), int(rng.integers(1, 120))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else \
                (np.concatenate([t[:k], t[k + L:]]) if len(t) - L > 1 else t)
        pairs.append((q, t))
        ws.append(int(rng.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 64, 100, 128, 200, 300])))
    _check_fast(solo_engine, oracle, pairs, ws)


def test_wave_kernel_config2_shape(solo_engine, oracle):
    """BASELINE config 2 shape (1000 x ~1000, w=128, 10 % divergence) + the other headline bands."""
    rng = np.random.default_rng(123)
    pairs, ws = [], []
    for w in (64, 128, 128, 128, 256):
        for _ in range(6):
            q = random_codes(rng, 1000)
            pairs.append((q, mutate(rng, q)))
            ws.append(w)
    _check_fast(solo_engine, oracle, pairs, ws)


def test_wave_kernel_other_scorings(solo_engine, oracle):
    rng = np.random.default_rng(321)
    for _ in range(40):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 70)), int(rng.integers(0, 6))
        pairs = []
        for _ in range(10):
            q = random_codes(rng, int(rng.integers(1, 400)), 0.03)
            d = rng.random() * 0.2
            pairs.append((q, mutate(rng, q, d, d / 3, d / 3)))
        w = int(rng.choice([3, 16, 40, 100]))
        _check_fast(solo_engine, oracle, pairs, [w] * len(pairs), mat=sedef_mat(ma, mi), gapo=go, gape=ge)


def test_wave_kernel_full_band_sedef_shapes(solo_engine, oracle):
    """w=-1 (every SEDEF call): gap fills, 500x500 side extensions, <=1000^2 gaps -> NREG 1..8."""
    rng = np.random.default_rng(2026)
    pairs = []
    for ql, tl in [(210, 209), (37, 3), (3, 180), (500, 500), (500, 431), (431, 500), (1000, 1000), (1000, 640),
                   (640, 1000), (977, 1000), (1, 1), (1, 300), (300, 1), (16, 16), (17, 15), (129, 127), (513, 511),
                   (1008, 1008), (1024, 990), (800, 120), (120, 800)]:
        for _ in range(2):
            q = random_codes(rng, ql, 0.01 if rng.random() < 0.3 else 0.0)
            t = mutate(rng, q, 0.05, 0.02, 0.02)
            t = t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])
            pairs.append((q, t))
    _check_fast(solo_engine, oracle, pairs, [-1] * len(pairs))


def _fnv(words):
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for w in words.tolist():
            h = (h ^ np.uint64(w)) * np.uint64(1099511628211)
    return int(h)


def test_config2_batch_checksums(engine, oracle):
    """BASELINE config 2 inputs (seed 42): per-task CIGAR checksums and scores of a 3000-task slice against
    the CPU path (reference kernel when oracle/_ref is present, else the oracle port), and size-independent
    properties on every task: CIGAR consumes exactly both sequences, counters add up."""
    import bench
    import sedef_amd
    from oracle.binding import Reference
    n = 3000
    pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, seed=42)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
    tasks["w"], tasks["zdrop"] = 128, -1
    res, cig = engine.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    try:
        cpu = Reference()
    except Exception:
        cpu = oracle
    score, h = cpu.batch(pool, q_off, qlen, t_off, tlen, w=128)
    assert np.array_equal(res["score"], score)
    for k in range(n):
        c = cig[int(res["cigar_off"][k]):int(res["cigar_off"][k]) + int(res["n_cigar"][k])]
        assert _fnv(c) == int(h[k]), k
        ops, lens = c & 0xf, c >> 4
        assert int(lens[ops != 2].sum()) == qlen[k] and int(lens[ops != 1].sum()) == tlen[k]
        assert int(res["matches"][k] + res["mismatches"][k]) == int(lens[ops == 0].sum())
        assert int(res["gap_bases"][k]) == int(lens[ops != 0].sum()) and int(res["gaps"][k]) == int((ops != 0).sum())


def test_general_and_wave_kernels_agree_at_scale(engine):
    """Idempotence / cross-kernel property at BASELINE sizes: the two independent DP kernels produce the same
    CIGARs and scores on 2000 config-2 tasks (want=ALL routes to the LDS kernel, want=CIGAR|SCORE to the wave kernel)."""
    import bench
    import sedef_amd
    n = 2000
    pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, seed=7)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
    tasks["w"], tasks["zdrop"] = 128, -1
    r1, c1 = engine.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_ALL)
    r2, c2 = engine.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    for f in ("score", "mte", "mte_q", "n_cigar", "cigar_off", "matches", "mismatches", "gaps", "gap_bases"):
        assert np.array_equal(r1[f], r2[f]), f
    assert np.array_equal(c1, c2)


def test_very_long_task_hbm_state(engine, oracle):
    """Longer than LDS can hold (> ~14k): the general kernel keeps its arena in an HBM slab."""
    rng = np.random.default_rng(31)
    q = random_codes(rng, 16500, 0.001)
    t = mutate(rng, q, 0.04, 0.01, 0.01)
    k = 7000
    t = np.concatenate([t[:k], random_codes(rng, 900), t[k:]])
    for w, want in ((-1, 7), (300, 7)):
        res, cig = engine.align_pairs([(q, t)], w=w, want=want)
        exp = oracle.extz2(q, t, w=w)
        got = _rec_to_dict(res[0], cig)
        for f in FIELDS:
            assert got[f] == exp[f], (f, w)
        assert cigar_to_str(got["cigar"]) == cigar_to_str(exp["cigar"])


def test_wave_kernel_fuzz_all_register_counts(solo_engine, oracle):
    """Random shapes up to 1100 with bands that map onto every NREG variant (and onto the general kernel
    beyond 1024 window cells), unrelated as well as related sequences."""
    rng = np.random.default_rng(4242)
    pairs, ws = [], []
    for _ in range(500):
        ql = int(rng.integers(1, 1100))
        q = random_codes(rng, ql, 0.01 if rng.random() < 0.2 else 0.0)
        if rng.random() < 0.25:
            t = random_codes(rng, int(rng.integers(1, 1100)))
        else:
            d = rng.random() * 0.1
            t = mutate(rng, q, d, d / 3, d / 3)
            if rng.random() < 0.3:
                k = int(rng.integers(0, len(t)))
                t = np.concatenate([t[:k], random_codes(rng, int(rng.integers(1, 300))), t[k:]])
        pairs.append((q, t))
        ws.append(int(rng.choice([-1, -1, 3, 20, 50, 120, 200, 260, 400, 480, 700])))
    _check_fast(solo_engine, oracle, pairs, ws)


def test_tiny_tasks_bulk(engine, oracle):
    """SEDEF's most frequent call shape: gap fills of a few bases (w=-1)."""
    rng = np.random.default_rng(99)
    pairs = []
    for _ in range(3000):
        q = random_codes(rng, int(rng.integers(1, 24)))
        t = random_codes(rng, int(rng.integers(1, 24))) if rng.random() < 0.5 else mutate(rng, q, 0.1, 0.05, 0.05)
        pairs.append((q, t))
    _check_fast(engine, oracle, pairs, [-1] * len(pairs))


# ---- lane kernel: one small full-band task per lane (extz2_lane.hip) ----
def _lane_pairs(rng, n):
    """Small tasks of every shape the lane kernel takes (both sequences <= 256 bases, <= 16,384 cells): gap fills of a few
    bases, the 210 x 209 limit of chain gaps, one-base sequences, N runs and all-N, unrelated sequences, homopolymers."""
    pairs = []
    for k in range(n):
        u = rng.random()
        if u < 0.4:
            ql, tl = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        elif u < 0.8:
            ql = int(rng.integers(5, 101))
            tl = max(1, min(209, ql + int(rng.integers(-20, 21))))
        elif u < 0.9:
            ql = int(rng.integers(1, 257))
            tl = int(rng.integers(1, min(256, 16384 // ql) + 1))
        else:
            ql, tl = [(1, 256), (256, 1), (64, 256), (256, 64), (128, 128), (210, 78), (33, 32), (17, 31), (8, 9), (1, 1)][k % 10]
        kind = rng.random()
        q = random_codes(rng, ql, 0.05 if kind < 0.2 else 0.0)
        if kind < 0.55:
            t = _fit(rng, mutate(rng, q, 0.06, 0.03, 0.03), tl)
        elif kind < 0.8:
            t = random_codes(rng, tl, 0.02)
        elif kind < 0.9:
            t = np.full(tl, int(rng.integers(0, 4)), np.uint8)
            q = np.full(ql, int(t[0]) if rng.random() < 0.5 else int(rng.integers(0, 4)), np.uint8)
        else:
            t = np.full(tl, 4, np.uint8) if rng.random() < 0.5 else _fit(rng, q, tl)
        pairs.append((q, t))
    return pairs


def _fit(rng, t, tl):
    return t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])


def test_lane_kernel_small_full_band_tasks(oracle):
    """A batch of small full-band tasks runs one task per LANE, sorted and planned on the device: every field the fast path
    returns, the CIGAR and the column counts against the oracle; explicit bands >= both lengths count as full; reversed
    CIGARs and score-only tasks in between."""
    eng = _engine_with_env(SDF_LANE_MIN=2048)
    rng = np.random.default_rng(7001)
    pairs = _lane_pairs(rng, 9000)
    ws = [-1 if k % 3 else max(len(q), len(t)) + k % 5 for k, (q, t) in enumerate(pairs)]
    flags = [0x80 if k % 11 == 3 else 0x01 if k % 13 == 5 else 0 for k in range(len(pairs))]
    _check_fast(eng, oracle, pairs, ws, flags)
    assert eng.last_lane_tasks() == len(pairs)
    # want = SCORE for the whole batch: no flags, no CIGAR
    import sedef_amd
    res, cig = eng.align_pairs(pairs[:3000], w=-1, want=sedef_amd.extz2.WANT_SCORE)
    assert eng.last_lane_tasks() == 3000 and len(cig) == 0
    for (q, t), r in zip(pairs[:3000], res):
        exp = oracle.extz2(q, t)
        assert (int(r["score"]), int(r["mte"]), int(r["mte_q"]), int(r["n_cigar"])) == (exp["score"], exp["mte"], exp["mte_q"], 0)


def test_lane_kernel_other_scorings(oracle):
    """Tame scorings (match + 2 (q + e) <= 127: every byte of the reference's state stays in 0..127) take the lane kernel and
    agree with the oracle; a scoring beyond that keeps to the window kernels, which emulate the wrap-around."""
    eng = _engine_with_env(SDF_LANE_MIN=1024)
    rng = np.random.default_rng(7002)
    took = 0
    for it in range(14):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 60)), int(rng.integers(0, 5))
        if it == 0:
            ma, mi, go, ge = 5, -4, 60, 1   # cap = 127: the last tame one
        if it == 1:
            ma, mi, go, ge = 5, -4, 61, 1   # cap = 129: not tame
        if it == 2:
            ma, mi, go, ge = 1, -1, 0, 0
        pairs = _lane_pairs(rng, 1500)
        _check_fast(eng, oracle, pairs, [-1] * len(pairs), mat=sedef_mat(ma, mi), gapo=go, gape=ge)
        tame = ma + 2 * (go + ge) <= 127 and -mi <= 2 * (go + ge)
        assert (eng.last_lane_tasks() == len(pairs)) == tame, (ma, mi, go, ge, eng.last_lane_tasks())
        took += tame
    assert took >= 8


def test_lane_kernel_next_to_the_other_kernels(oracle):
    """One batch with lane tasks, banded tasks of the same sizes (not eligible), wide full-band tasks (stripe kernel) and
    long banded ones: the lane tasks leave the chunks, everything else is planned as before, one result array."""
    eng = _engine_with_env(SDF_LANE_MIN=1024)
    rng = np.random.default_rng(7003)
    pairs = _lane_pairs(rng, 6000)
    ws = [-1] * len(pairs)
    for k in range(0, len(pairs), 7):  # a band below the lengths: not a lane task
        ws[k] = max(1, max(len(pairs[k][0]), len(pairs[k][1])) // 2)
    extra, extra_w = _stripe_mix(rng)
    for k, (p_, w_) in enumerate(zip(extra, extra_w)):
        pairs.insert(500 * k + 17, p_)
        ws.insert(500 * k + 17, w_)
    _check_dropped_max(eng, oracle, pairs, ws)
    n_lane = sum(1 for (q, t), w in zip(pairs, ws) if len(q) <= 256 and len(t) <= 256 and len(q) * len(t) <= 16384 and
                 (w < 0 or w >= max(len(q), len(t))))
    assert eng.last_lane_tasks() == n_lane > 4000


# ---- strip kernel: full-band tasks of 257..1024 target bases, row-major strips, two tasks per wavefront (extz2_strip.hip) ----
def _strip_pairs(rng, n):
    """Tasks the strip kernel takes -- every block count and ragged last strips, queries from 64 bases to several
    thousand, partners of equal geometry and tasks without one, N runs, unrelated sequences, big indels."""
    pairs = []
    geos = [(500, 500), (500, 431), (500, 377), (64, 257), (1000, 1000), (1000, 1024), (777, 512), (300, 513), (2500, 600),
            (128, 1023), (999, 264), (65, 300)]
    for k in range(n):
        ql, tl = geos[k % len(geos)] if k % 3 else (int(rng.integers(64, 1500)), int(rng.integers(257, 1025)))
        kind = rng.random()
        q = random_codes(rng, ql, 0.01 if kind < 0.25 else 0.0)
        if kind < 0.7:
            t = mutate(rng, q, 0.06, 0.02, 0.02)
            if kind < 0.3 and len(t) > 300:
                at, ln = int(rng.integers(0, len(t) - 100)), int(rng.integers(1, 300))
                t = np.concatenate([t[:at], random_codes(rng, ln), t[at:]]) if rng.random() < 0.5 else np.concatenate([t[:at], t[at + ln:]])
            t = _fit(rng, t, tl)
        elif kind < 0.9:
            t = random_codes(rng, tl, 0.02)
        else:
            t = np.full(tl, 4, np.uint8) if rng.random() < 0.3 else np.full(tl, int(q[0]) & 3, np.uint8)
        pairs.append((q, t))
    return pairs


@pytest.fixture(scope="module")
def strip_engine():
    """The strip kernels whatever the number of tasks (they take a chunk's tasks only when there are enough to fill the
    device: SDF_STRIP_ALWAYS lifts that for the parity tests)."""
    return _engine_with_env(SDF_STRIP_ALWAYS=1)


def test_strip_kernel_full_band_mid_size_tasks(strip_engine, oracle):
    engine = strip_engine
    rng = np.random.default_rng(8001)
    pairs = _strip_pairs(rng, 400)
    ws = [-1 if k % 4 else max(len(q), len(t)) for k, (q, t) in enumerate(pairs)]
    flags = [0x80 if k % 9 == 4 else 0x01 if k % 11 == 6 else 0 for k in range(len(pairs))]
    _check_fast(engine, oracle, pairs, ws, flags)
    assert engine.last_paired() >= 100  # partners of equal geometry share a wavefront
    assert engine.last_reran() == 0
    # the same tasks on the window / stripe kernels
    old = _engine_with_env(SDF_NO_STRIP=1)
    _check_fast(old, oracle, pairs[:60], ws[:60], flags[:60])


def _wide_strip_pairs(rng):
    """Full-band tasks whose targets take 3 .. 16 blocks of 512 columns: a chain of wavefronts (extz2_strip.hip)."""
    pairs = []
    for ql, tl in [(1500, 1025), (1500, 1030), (3000, 3000), (2900, 3050), (6000, 6000), (700, 8192), (5000, 2049),
                   (2000, 1537), (2000, 1537), (2100, 1600), (64, 4000), (1200, 1200), (1201, 1234), (4097, 4100)]:
        kind = rng.random()
        q = random_codes(rng, ql, 0.004 if kind < 0.3 else 0.0)
        if kind < 0.8:
            t = mutate(rng, q, 0.06, 0.02, 0.02)
            at = int(rng.integers(0, max(1, len(t) - 1)))
            t = np.concatenate([t[:at], random_codes(rng, int(rng.integers(1, 400))), t[at:]])
            t = _fit(rng, t, tl)
        else:
            t = random_codes(rng, tl)
        pairs.append((q, t))
    return pairs


def test_strip_chain_wide_full_band_tasks(strip_engine, oracle):
    engine = strip_engine
    rng = np.random.default_rng(8003)
    pairs = _wide_strip_pairs(rng)
    flags = [0x80 if k % 5 == 2 else 0x01 if k == 7 else 0 for k in range(len(pairs))]
    _check_fast(engine, oracle, pairs, [-1] * len(pairs), flags)
    assert engine.last_paired() >= 8 and engine.last_reran() == 0
    # between tasks of the other kernels, several chains side by side
    more = _strip_pairs(rng, 30) + pairs + _wide_strip_pairs(rng)
    _check_fast(engine, oracle, more, [-1] * len(more))
    # a wait that gives up: both tasks of the pair run again on another kernel
    eng = _engine_with_env(SDF_STRIPE_SPIN_CAP=1, SDF_STRIP_ALWAYS=1)
    _check_fast(eng, oracle, pairs, [-1] * len(pairs), flags)
    assert eng.last_reran() >= 6
    # four columns per lane (blocks of 256 columns: twice as many, shorter steps: what a chunk of few chains takes by itself
    # since round 4) and eight (a chunk of more than 4,096 chain wavefronts)
    for cols in (4, 8):
        engc = _engine_with_env(SDF_STRIP_COLS=cols, SDF_STRIP_ALWAYS=1)
        _check_fast(engc, oracle, pairs, [-1] * len(pairs), flags)
    # by itself the planner gives so few wide tasks to the stripe kernel
    _check_fast(_engine_with_env(SDF_NO_LANE=1), oracle, pairs[:6], [-1] * 6)


def test_strip_chain_targets_wider_than_8192(engine, strip_engine, oracle):
    """Full-band targets beyond 8192 bases (the stage's far-gap tasks: 8-10 kb a side and up to the 60 kb of a chunk,
    src/align.cc:129-175): a few of them take the stripe kernel (up to 254 stripes: 32,512 bases), many -- or wider ones -- chains
    of up to 256 blocks; before round 4 they were the workgroup kernel's, ~3 us per row."""
    rng = np.random.default_rng(8004)
    pairs = []
    for ql, tl in [(9000, 8200), (8700, 8700), (8650, 8710), (3000, 12000), (12000, 9000), (20000, 16400), (64, 30000),
                   (15000, 20000), (500, 8193)]:
        q = random_codes(rng, ql, 0.003 if len(pairs) % 3 == 1 else 0.0)
        t = mutate(rng, q, 0.05, 0.01, 0.01)
        at = int(rng.integers(0, max(1, len(t) - 1)))
        t = _fit(rng, np.concatenate([t[:at], random_codes(rng, int(rng.integers(1, 3000))), t[at:]]), tl)
        pairs.append((q, t))
    flags = [0x80 if k == 3 else 0x01 if k == 6 else 0 for k in range(len(pairs))]
    _check_fast(engine, oracle, pairs, [-1] * len(pairs), flags)  # (nine tasks: the stripe kernel)
    assert engine.last_reran() == 0
    _check_fast(strip_engine, oracle, pairs, [-1] * len(pairs), flags)  # (chains whatever their number)
    assert strip_engine.last_reran() == 0
    # chains of eight columns per lane (blocks of 512 columns: few chains take four by themselves) and with a wait that gives up
    _check_fast(_engine_with_env(SDF_STRIP_COLS=8, SDF_STRIP_ALWAYS=1), oracle, pairs[:5], [-1] * 5, flags[:5])
    for env in ({"SDF_STRIP_ALWAYS": 1}, {}):
        eng = _engine_with_env(SDF_STRIPE_SPIN_CAP=1, **env)
        _check_fast(eng, oracle, pairs[:4], [-1] * 4, flags[:4])
        assert eng.last_reran() >= 2


def test_strip_chain_one_task_of_a_whole_chunk(engine, oracle):
    """The largest task the stage can ask for: both sides at the chunk length of align_helper's loop (MAX_KSW_SEQ_LEN = 60 kb,
    src/align.cc:46-57), full band -- 240 chain blocks of 256 columns, 3.7e9 cells, 3.7 GB of direction flags."""
    rng = np.random.default_rng(8005)
    q = random_codes(rng, 61440, 0.001)
    t = mutate(rng, q, 0.04, 0.008, 0.008)
    t = _fit(rng, np.concatenate([t[:20000], random_codes(rng, 1500), t[20000:]]), 61440)
    _check_fast(engine, oracle, [(q, t)], [-1])
    assert engine.last_reran() == 0


def test_strip_kernel_other_scorings(strip_engine, oracle):
    engine = strip_engine
    rng = np.random.default_rng(8002)
    for it in range(10):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 60)), int(rng.integers(0, 5))
        if it == 0:
            ma, mi, go, ge = 5, -4, 60, 1  # cap = 127
        if it == 1:
            ma, mi, go, ge = 5, -4, 61, 1  # not tame: the stripe kernels
        if it == 2:
            ma, mi, go, ge = 5, -10, 10, 0  # mismatch + q + 2 e = 0: the last scoring whose z - q the strips may take with a 32-bit subtract
        if it == 3:
            ma, mi, go, ge = 5, -11, 10, 0  # ... and the first that keeps to the window kernels (tame, but z - q can be -1)
        pairs = _strip_pairs(rng, 36)
        _check_fast(engine, oracle, pairs, [-1] * len(pairs), mat=sedef_mat(ma, mi), gapo=go, gape=ge)


def _same_geometry_tasks(rng, ql, tl, copies, n_frac=0.0):
    """`copies` tasks with the same lengths and unrelated contents (related, diverged, random, N runs)."""
    out = []
    for _ in range(copies):
        q = random_codes(rng, ql, n_frac if rng.random() < 0.4 else 0.0)
        kind = rng.random()
        if kind < 0.2:
            t = random_codes(rng, tl)
        else:
            d = rng.random() * 0.15
            t = mutate(rng, q, d, d / 3, d / 3)
            if kind > 0.7 and len(t) > 4:
                k, L = int(rng.integers(0, len(t))), int(rng.integers(1, 120))
                t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else \
                    np.concatenate([t[:k], t[k + L:]])
            t = _fit(rng, t, tl)
        out.append((q, t))
    return out


def test_pair_kernel_fuzz_banded(engine, oracle):
    rng = np.random.default_rng(777)
    pairs, ws = [], []
    for _ in range(450):
        ql = int(rng.integers(1, 900))
        tl = max(1, ql + int(rng.integers(-60, 60)))
        w = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 64, 100, 128, 200, 300]))
        copies = int(rng.choice([2, 2, 3, 4]))
        pairs += _same_geometry_tasks(rng, ql, tl, copies, 0.02)
        ws += [w] * copies
    perm = rng.permutation(len(pairs))  # partners are found by the planner, not by adjacency
    pairs, ws = [pairs[i] for i in perm], [ws[i] for i in perm]
    _check_fast(engine, oracle, pairs, ws)
    assert engine.last_paired() >= 600  # narrow bands that cannot reach the corner go to the general kernel


def test_pair_kernel_all_register_counts(engine, oracle):
    """Windows of 16..512 slots -> 1, 2, 3, 4, 6, 8 registers; full band and banded; wider windows and odd
    tasks out stay on the one-task wave kernel inside the same batch (full-band tasks of more than 400 target
    bases: the stripe kernel)."""
    rng = np.random.default_rng(778)
    pairs, ws = [], []
    for ql, tl, w in [(8, 8, -1), (30, 30, -1), (60, 50, -1), (100, 120, -1), (150, 150, -1), (210, 209, -1),
                      (300, 280, -1), (390, 395, -1), (400, 399, -1), (400, 410, -1), (500, 500, -1), (480, 500, -1), (500, 431, -1),
                      (1000, 1000, 20), (1000, 990, 60), (1000, 1010, 100), (1000, 1003, 128), (990, 1000, 128),
                      (1000, 1000, 160), (1000, 980, 200), (1000, 1000, 260), (1000, 1000, 350), (900, 1000, 440),
                      (1000, 1000, 470), (1000, 1000, 500), (700, 700, -1), (1, 1, -1), (1, 40, -1), (40, 1, -1),
                      (16, 16, 3), (17, 15, 1), (2000, 2000, 128), (3000, 2990, 64)]:
        copies = 3
        pairs += _same_geometry_tasks(rng, ql, tl, copies, 0.01)
        ws += [w] * copies
    _check_fast(engine, oracle, pairs, ws)
    assert engine.last_paired() >= 44


def test_pair_kernel_other_scorings(engine, oracle):
    rng = np.random.default_rng(779)
    for _ in range(40):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 70)), int(rng.integers(0, 6))
        pairs, ws = [], []
        for _ in range(5):
            ql = int(rng.integers(1, 400))
            pairs += _same_geometry_tasks(rng, ql, max(1, ql + int(rng.integers(-20, 20))), 2, 0.03)
            ws += [int(rng.choice([3, 16, 40, 100, -1]))] * 2
        _check_fast(engine, oracle, pairs, ws, mat=sedef_mat(ma, mi), gapo=go, gape=ge)


def test_pair_kernel_golden_doubled(engine, golden_cases):
    """Every golden case twice in one batch: each meets its twin in a wavefront; both halves must be right."""
    import sedef_amd
    sel = [c for c in golden_cases if (c["flag"] & ~0x81) == 0 and c["zdrop"] < 0
           and (c["match"], c["mismatch"], c["gapo"], c["gape"]) == (5, -4, 40, 1)]
    sel = sel + sel
    pairs = [(codes(c["q"]), codes(c["t"])) for c in sel]
    res, cig = engine.align_pairs(pairs, w=[c["w"] for c in sel], flag=[c["flag"] for c in sel],
                                  want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    assert engine.last_paired() >= len(sel) // 2
    for c, r in zip(sel, res):
        exp = c["expect"]
        got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        for fld in FAST_FIELDS:
            assert int(r[fld]) == exp[fld], (c["tag"], fld, c["w"], len(c["q"]), len(c["t"]))
        assert cigar_to_str(got) == exp["cigar"], (c["tag"], c["w"], len(c["q"]), len(c["t"]))


def test_pair_and_wave_kernels_agree_at_scale(engine, solo_engine):
    """20,000 config-2 tasks: the batch with pairing (default) and with SDF_NO_PAIR=1 give the same bytes."""
    import sedef_amd
    import bench
    n = 20000
    pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, 4321)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
    tasks["w"], tasks["zdrop"] = 128, -1
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    res1, cig1 = engine.align_batch(tasks, pool, want=want)
    assert engine.last_paired() > 15000
    solo = solo_engine
    res2, cig2 = solo.align_batch(tasks, pool, want=want)
    assert solo.last_paired() == 0
    assert res1.tobytes() == res2.tobytes()
    assert np.array_equal(cig1, cig2)


def test_pair_kernel_self_pairs(engine, oracle, golden_cases):
    """Tasks without a same-geometry partner run as a pair with themselves: the one-task fuzz sets again, on the
    default context."""
    test_wave_kernel_golden(engine, golden_cases)
    test_wave_kernel_fuzz_banded(engine, oracle)
    test_wave_kernel_full_band_sedef_shapes(engine, oracle)
    test_wave_kernel_fuzz_all_register_counts(engine, oracle)


# ---- BASELINE configs[3] / configs[4] shapes as parity cases (the bench line is configs[1]) ----
def _batch_vs_cpu(engine, oracle, batch, w):
    """Every task of the batch through the product path (CIGAR + score + counts) against the CPU path: scores,
    CIGAR checksums, and the CIGAR consuming exactly both sequences."""
    import sedef_amd
    from oracle.binding import Reference
    pool, q_off, qlen, t_off, tlen = batch
    n = len(qlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
    tasks["w"], tasks["zdrop"] = w, -1
    res, cig = engine.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    try:
        cpu = Reference()
    except Exception:
        cpu = oracle
    ws = np.broadcast_to(np.asarray(w, np.int32), (n,))
    for wv in np.unique(ws):
        sel = np.nonzero(ws == wv)[0]
        score, h = cpu.batch(pool, q_off[sel], qlen[sel], t_off[sel], tlen[sel], w=int(wv))
        assert np.array_equal(res["score"][sel], score), int(wv)
        for k, hk in zip(sel, h):
            c = cig[int(res["cigar_off"][k]):int(res["cigar_off"][k]) + int(res["n_cigar"][k])]
            assert _fnv(c) == int(hk), (int(k), int(qlen[k]), int(tlen[k]), int(wv))
            if len(c) and not int(res["zdropped"][k]):  # (a band that cannot reach the corner stops early)
                ops, lens = c & 0xf, c >> 4
                assert int(lens[ops != 2].sum()) == qlen[k] and int(lens[ops != 1].sum()) == tlen[k]
                assert int(res["matches"][k] + res["mismatches"][k]) == int(lens[ops == 0].sum())
    return res


def test_config4_hg19_task_mixture(engine, oracle):
    """configs[3]: hg19-shaped task-size mixture, w=-1 (SEDEF's real mode): tiny gap fills, 500x500 side
    extensions, a few tasks beyond 1e6 cells -- every kernel and the planner's pairing in one batch."""
    import bench
    batch, w = bench.synth_hg19_mixture(30000, seed=404, big=2500)
    _batch_vs_cpu(engine, oracle, batch, w)
    assert engine.last_lane_tasks() > 29000  # the gap fills, one per lane (extz2_lane.hip); the rest in pairs and stripes
    solo = _engine_with_env(SDF_NO_LANE=1)
    _batch_vs_cpu(solo, oracle, batch, w)
    assert solo.last_lane_tasks() == 0 and solo.last_paired() > 20000
    # the batch in two parts (SDF_SPLIT_MIN: the first eighth on a second context while the rest is still being read)
    split = _engine_with_env(SDF_SPLIT_MIN=20000)
    _batch_vs_cpu(split, oracle, batch, w)
    # (the first part -- an eighth of the batch, fewer tasks than the lane kernel asks for -- runs on the window kernels of
    # the part's own context; until round 5 a function-local static froze the first SDF_SPLIT_MIN a process saw and this
    # context never split)
    assert 24000 < split.last_lane_tasks() < 27000


def test_config5_mm8_mixed_bands(engine, oracle):
    """configs[4]: lengths log-uniform 200..20,000, bands 64..512 mixed inside one batch, one-sided kb-scale
    indels (band cannot reach the corner -> the reference's early stop, zdropped / empty CIGAR, must match)."""
    import bench
    batch, w = bench.synth_mm8_mixture(240, seed=505, max_len=20000)
    _batch_vs_cpu(engine, oracle, batch, w)


def test_plain_general_kernel_wide_windows(engine, oracle):
    """Windows beyond 1024 cells with only CIGAR / score / mte wanted: the PLAIN flavour of the general kernel
    (packed recurrence, H along the band edge) in its 256- and 1024-thread, LDS and HBM-state instantiations,
    against the oracle; full band and wide bands, N runs, big indels, unrelated sequences."""
    rng = np.random.default_rng(9090)
    pairs, ws = [], []
    for ql, tl, w in [(1100, 1100, -1), (1500, 1300, -1), (2100, 2300, -1), (3000, 3000, -1), (1025, 4000, -1),
                      (4000, 1040, -1), (2500, 2500, 700), (3000, 2800, 1100), (6000, 5800, 1500), (5000, 5000, -1),
                      (1200, 1200, 600), (2000, 2100, 520)]:
        for kind in range(2):
            q = random_codes(rng, ql, 0.005 if kind else 0.0)
            if kind and ql > 2000:
                t = random_codes(rng, tl)  # unrelated
            else:
                t = mutate(rng, q, 0.06, 0.02, 0.02)
                k = int(rng.integers(0, len(t)))
                t = np.concatenate([t[:k], random_codes(rng, int(rng.integers(1, 300))), t[k:]])
                t = _fit(rng, t, tl)
            pairs.append((q, t))
            ws.append(w)
    _check_fast(engine, oracle, pairs, ws)
    # HBM-state instantiation (> ~14k bases)
    q = random_codes(rng, 15500, 0.001)
    t = mutate(rng, q, 0.04, 0.01, 0.01)
    t = np.concatenate([t[:6000], random_codes(rng, 700), t[6000:]])
    _check_fast(engine, oracle, [(q, t), (q[:15000], t[:15100])], [-1, 1400])


def test_pair_kernel_long_sequences_streamed_windows(onetask_engine, oracle):
    engine = onetask_engine
    """Sequences far longer than the LDS windows of the pair kernel (window slots + 1024 entries): the target and
    reversed-query windows are re-filled from the packed pool as the band moves; N runs and indels near the band
    edge included; several same-geometry tasks per shape so that pairs form."""
    rng = np.random.default_rng(6060)
    pairs, ws = [], []
    for ql, tl, w in [(3000, 3000, 64), (5000, 5010, 100), (9000, 8990, 128), (20000, 20000, 200), (12000, 12100, 300),
                      (2500, 2400, 440), (1500, 1500, 17), (30000, 30020, 64)]:
        for _ in range(3):
            q = random_codes(rng, ql, 0.003 if rng.random() < 0.5 else 0.0)
            d = rng.random() * 0.08
            t = mutate(rng, q, d, d / 4, d / 4)
            if rng.random() < 0.5 and len(t) > 200:
                k, L = int(rng.integers(0, len(t) - 100)), int(rng.integers(1, max(2, w // 2)))
                t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else \
                    np.concatenate([t[:k], t[k + L:]])
            pairs.append((q, _fit(rng, t, tl)))
            ws.append(w)
    _check_fast(engine, oracle, pairs, ws)
    assert engine.last_paired() >= 14


def test_wave_kernel_long_sequences_streamed_windows(onetask_engine, onetask_solo_engine, oracle):
    engine, solo_engine = onetask_engine, onetask_solo_engine
    """Long sequences through the one-task-per-wavefront kernel: its LDS sequence windows are re-filled as the band
    moves (bands of 513..1000 cells take this kernel in the default context too; narrower ones with SDF_NO_PAIR)."""
    rng = np.random.default_rng(7070)
    pairs, ws = [], []
    for ql, tl, w in [(4000, 4000, 64), (9000, 9050, 128), (20000, 19990, 300), (20000, 20000, 512), (8000, 8000, 700),
                      (6000, 6100, 990), (2500, 2500, 17), (25000, 25000, 40)]:
        for _ in range(2):
            q = random_codes(rng, ql, 0.003 if rng.random() < 0.5 else 0.0)
            d = rng.random() * 0.08
            t = mutate(rng, q, d, d / 4, d / 4)
            if rng.random() < 0.5:
                k, L = int(rng.integers(0, len(t) - 100)), int(rng.integers(1, max(2, w // 2)))
                t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else \
                    np.concatenate([t[:k], t[k + L:]])
            pairs.append((q, _fit(rng, t, tl)))
            ws.append(w)
    _check_fast(solo_engine, oracle, pairs, ws)
    _check_fast(engine, oracle, pairs, ws)


def test_stripe_kernel_wide_full_band(engine, oracle):
    """Full-band tasks with targets of 1025..8192 positions: a workgroup of wavefronts, one per target stripe
    (extz2_stripe.hip: 128 / 256 / 512-position stripes), against the oracle -- every stripe count incl. ragged last
    stripes, short and long queries, N runs, unrelated sequences, big indels."""
    rng = np.random.default_rng(8181)
    pairs = []
    for ql, tl in [(1100, 1100), (1500, 1025), (900, 1300), (2048, 2048), (2100, 2049), (300, 2000), (3000, 3000),
                   (4000, 4096), (4097, 4100), (6000, 6000), (1200, 8192), (5000, 7000), (40, 1500), (2500, 1026)]:
        for kind in range(2):
            q = random_codes(rng, ql, 0.004 if kind else 0.0)
            if kind and ql > 2000:
                t = random_codes(rng, tl)
            else:
                t = mutate(rng, q, 0.06, 0.02, 0.02)
                k = int(rng.integers(0, max(1, len(t) - 1)))
                t = np.concatenate([t[:k], random_codes(rng, int(rng.integers(1, 400))), t[k:]])
                t = _fit(rng, t, tl)
            pairs.append((q, t))
    _check_fast(engine, oracle, pairs, [-1] * len(pairs))


def _check_dropped_max(engine, oracle, pairs, ws):
    """_check_fast, plus the best cell (max, max_t, max_q) of the tasks whose band runs out: the traceback starts there."""
    import sedef_amd
    _check_fast(engine, oracle, pairs, ws)
    res, _ = engine.align_pairs(pairs, w=ws, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    ndrop = 0
    for (q, t), w, r in zip(pairs, ws, res):
        exp = oracle.extz2(q, t, w=w)
        if exp["zdropped"]:
            ndrop += 1
            assert (int(r["max"]), int(r["max_t"]), int(r["max_q"])) == (exp["max"], exp["max_t"], exp["max_q"]), (w, len(q), len(t))
    return ndrop


@pytest.mark.parametrize("nreg", [1, 2, 4])
def test_stripe_kernels_every_width_fuzz(oracle, nreg):
    """Both stripe kernels with the stripe width forced (128 / 256 / 512 positions) and every banded / full-band task
    of 100+ anti-diagonals routed to them: full band (extz2_stripe.hip) from 130 target bases, bands of 1..300 that
    reach the corner or run out (extz2_bstripe.hip), N bases, indels longer than the band, one-base sequences."""
    eng = _engine_with_env(SDF_STRIPE_NREG=nreg, SDF_BSTRIPE_NREG=nreg if nreg > 1 else 0, SDF_STRIPE_MIN=130,
                           SDF_BSTRIPE_MIN_ROWS=100, SDF_BSTRIPE_ALL=1)
    rng = np.random.default_rng(9900 + nreg)
    pairs, ws = [], []
    for _ in range(260):
        ql = int(rng.integers(60, 1500))
        tl = max(1, ql + int(rng.integers(-250, 250)))
        w = int(rng.choice([-1, -1, 1, 2, 7, 15, 16, 17, 31, 33, 64, 100, 128, 200, 300]))
        q = random_codes(rng, ql, 0.004 if rng.random() < 0.3 else 0.0)
        t = mutate(rng, q, 0.05, 0.01, 0.01)
        if rng.random() < 0.4 and len(t) > 300:
            k, L = int(rng.integers(0, len(t) - 100)), int(rng.integers(1, 400))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else np.concatenate([t[:k], t[k + L:]])
        pairs.append((q, _fit(rng, t, tl)))
        ws.append(w)
    for ql, tl, w in [(1, 700, 5), (700, 1, 5), (2, 900, -1), (900, 2, -1), (129, 129, 64), (128, 2000, 100), (2000, 128, 100)]:
        q = random_codes(rng, ql)
        pairs.append((q, _fit(rng, mutate(rng, q, 0.05, 0.01, 0.01), tl)))
        ws.append(w)
    assert _check_dropped_max(eng, oracle, pairs, ws) >= 20


def test_stripe_kernels_other_scorings(oracle):
    """Both stripe kernels under other match / mismatch / gap scores (the CLI's --match ... overrides, src/align_main.cc:
    343-352): saturation of the difference bytes, gap-open 0, scores the reference clamps."""
    eng = _engine_with_env(SDF_STRIPE_MIN=130, SDF_BSTRIPE_MIN_ROWS=150, SDF_BSTRIPE_ALL=1)
    rng = np.random.default_rng(9933)
    for _ in range(36):
        ma, mi = int(rng.integers(1, 12)), -int(rng.integers(1, 12))
        go, ge = int(rng.integers(0, 70)), int(rng.integers(0, 6))
        pairs, ws = [], []
        for _ in range(10):
            ql = int(rng.integers(80, 900))
            tl = max(1, ql + int(rng.integers(-150, 150)))
            q = random_codes(rng, ql, 0.003 if rng.random() < 0.3 else 0.0)
            pairs.append((q, _fit(rng, mutate(rng, q, 0.06, 0.015, 0.015), tl)))
            ws.append(int(rng.choice([-1, -1, 3, 16, 40, 100, 200])))
        _check_fast(eng, oracle, pairs, ws, mat=sedef_mat(ma, mi), gapo=go, gape=ge)


def test_banded_stripe_kernel_score_refresh_reaches_the_next_stripe(oracle):
    """The reference refreshes scores in 16-cell strides from the band START, up to fifteen cells past its last computed
    block -- into the first columns of the next target stripe before that stripe computes anything; a cell computed later
    as part of a widened block still holds that score.  Found by fuzzing (279 x 460, w = 200: score -373 instead of -377):
    the committed pair, its prefixes (the alignment of the band start with the 16-cell blocks decides) and random pairs of
    that geometry."""
    import os
    eng = _engine_with_env(SDF_BSTRIPE_MIN_ROWS=200, SDF_BSTRIPE_ALL=1)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "bstripe_refresh_spill.npz"))
    q, t, w = d["q"], d["t"], int(d["w"])
    pairs = [(q[:n], t) for n in range(255, 280)] + [(q, t[:n]) for n in range(441, 461)]
    ws = [w] * len(pairs)
    rng = np.random.default_rng(9922)
    for _ in range(300):
        ww = int(rng.choice([100, 150, 176, 200, 232, 300]))
        ql = int(rng.integers(130, 700))
        tl = ql + ww - int(rng.integers(0, 40))
        qq = random_codes(rng, ql)
        pairs.append((qq, _fit(rng, mutate(rng, qq, 0.05, 0.01, 0.01), tl)))
        ws.append(ww)
    _check_fast(eng, oracle, pairs, ws)


def test_banded_stripe_kernel_long_tasks(engine, oracle):
    """Long banded tasks as the default context routes them (4000+ anti-diagonals, windows of more than 192 slots, or
    bands that run out): 128-position stripes up to 32 kb targets, one row block apart; bands that reach the corner, run
    out after a long indel, or never reach it at all."""
    rng = np.random.default_rng(9911)
    pairs, ws = [], []
    for ql, tl, w, indel in [(7000, 7000, 128, 0), (6500, 6400, 256, 0), (8000, 8000, 512, 300), (12000, 11000, 300, 0),
                             (7000, 9000, 512, 0), (20000, 20000, 200, 150), (3000, 3100, 100, 0), (9000, 9000, 300, 700),
                             (16200, 16300, 128, 0), (17000, 16500, 256, 0), (2100, 2000, 400, 0), (25000, 25010, 96, 40)]:
        q = random_codes(rng, ql, 0.002 if rng.random() < 0.5 else 0.0)
        t = mutate(rng, q, 0.05, 0.01, 0.01)
        if indel:
            k = int(rng.integers(0, len(t) - indel))
            t = np.concatenate([t[:k], t[k + indel:]])
        pairs.append((q, _fit(rng, t, tl)))
        ws.append(w)
    assert _check_dropped_max(engine, oracle, pairs, ws) >= 3


def _stripe_mix(rng):
    """Tasks of both stripe kernels (wide full band; long banded, some whose band runs out) between ordinary ones."""
    pairs, ws = [], []
    for ql, tl, w, indel in [(1500, 1400, -1, 0), (700, 700, 64, 0), (7000, 7000, 128, 0), (2500, 3000, -1, 0),
                             (300, 310, -1, 0), (8000, 8000, 512, 300), (900, 2000, -1, 0), (6500, 6400, 256, 0),
                             (1000, 1000, 128, 0), (4100, 4000, -1, 0), (9000, 9000, 300, 700), (50, 60, -1, 0)]:
        q = random_codes(rng, ql, 0.002 if rng.random() < 0.5 else 0.0)
        t = mutate(rng, q, 0.05, 0.01, 0.01)
        if indel:
            k = int(rng.integers(0, len(t) - indel))
            t = np.concatenate([t[:k], t[k + indel:]])
        pairs.append((q, _fit(rng, t, tl)))
        ws.append(w)
    return pairs, ws


def test_stripe_kernels_score_only(engine, oracle):
    """The stripe kernels keep their progress words, hand-over values and edge columns in HBM behind a task's direction
    flags: a task that wants no CIGAR still needs that room.  (a) a whole batch with want = SCORE, (b) SDF_FLAG_SCORE_ONLY
    on every other task of a CIGAR batch -- the neighbours' CIGARs must not be touched -- on both layouts."""
    import sedef_amd
    rng = np.random.default_rng(4242)
    pairs, ws = _stripe_mix(rng)
    pairs, ws = pairs * 3, ws * 3  # (several stripe tasks per launch: their sync regions must be disjoint)
    exp = [oracle.extz2(q, t, w=w) for (q, t), w in zip(pairs, ws)]
    res, cig = engine.align_pairs(pairs, w=ws, want=sedef_amd.extz2.WANT_SCORE)
    assert len(cig) == 0 and engine.last_reran() == 0
    for r, e, w in zip(res, exp, ws):
        assert (int(r["score"]), int(r["mte"]), int(r["zdropped"]), int(r["n_cigar"])) == (e["score"], e["mte"], e["zdropped"], 0), w
    SCORE_ONLY = 0x01  # KSW_EZ_SCORE_ONLY (extern/ksw2.h:10)
    flags = [SCORE_ONLY if k % 2 else 0 for k in range(len(pairs))]
    for shift in (0, 1):
        fl = flags[shift:] + flags[:shift]
        res, cig = engine.align_pairs(pairs, w=ws, flag=fl, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
        for r, e, w, f, (q, t) in zip(res, exp, ws, fl, pairs):
            assert (int(r["score"]), int(r["mte"]), int(r["zdropped"])) == (e["score"], e["mte"], e["zdropped"]), (w, f)
            got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
            if f:
                assert len(got) == 0
            else:
                assert cigar_to_str(got) == cigar_to_str(e["cigar"]), (w, len(q), len(t))


def test_stripe_wait_gives_up_and_tasks_run_again(oracle):
    """A stripe that stops waiting for its left neighbour (SDF_STRIPE_SPIN_CAP polls; the protocol's progress rests on
    the dispatch order) abandons its TASK; the batch call runs abandoned tasks again on the one-wavefront / one-workgroup
    kernels inside the same call.  Forced here with a cap of one poll: parity of the merged batch, CIGARs included, with
    reversed CIGARs and score-only tasks among the abandoned ones, next to tasks that never waited."""
    import sedef_amd
    eng = _engine_with_env(SDF_STRIPE_SPIN_CAP=1)
    rng = np.random.default_rng(4343)
    pairs, ws = _stripe_mix(rng)
    pairs, ws = pairs * 2, ws * 2
    REV = 0x80  # KSW_EZ_REV_CIGAR
    flags = [(REV if k % 3 == 1 else 0) | (0x01 if k % 7 == 5 else 0) for k in range(len(pairs))]
    want = sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE
    res, cig = eng.align_pairs(pairs, w=ws, flag=flags, want=want)
    assert eng.last_reran() >= 6
    for (q, t), w, f, r in zip(pairs, ws, flags, res):
        e = oracle.extz2(q, t, w=w, flag=f)
        assert (int(r["score"]), int(r["mte"]), int(r["zdropped"])) == (e["score"], e["mte"], e["zdropped"]), (w, f)
        got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        assert cigar_to_str(got) == cigar_to_str(e["cigar"]), (w, f, len(q), len(t))
    # the same context again, a larger batch (heavy split, several chunks): the counters of a call start at zero
    pairs2, ws2 = pairs * 40, ws * 40
    res2, cig2 = eng.align_pairs(pairs2, w=ws2, want=want)
    assert 100 <= eng.last_reran() <= 1024  # (the give-up list holds 65,536 tasks; beyond that the call fails)
    exp = [oracle.extz2(q, t, w=w) for (q, t), w in zip(pairs, ws)]
    for k, r in enumerate(res2):
        e = exp[k % len(pairs)]
        got = cig2[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        assert int(r["score"]) == e["score"] and cigar_to_str(got) == cigar_to_str(e["cigar"]), k


def test_pair_kernel_track_band_runs_out(engine, oracle):
    """Banded tasks whose band cannot reach the end of both sequences (|qlen - tlen| > w): the reference stops when
    the band is exhausted and backtracks from the best cell (extern/ksw2_extz2_sse.cc:116, :292-295).  The pair kernel's
    TRACK flavour follows the exact H of every cell for that: zdropped, max / max_t / max_q and the CIGAR against the
    oracle, for windows of 192 / 384 slots (wider ones: the general kernel), short and streamed sequences, ties
    (repeats), N runs."""
    import sedef_amd
    rng = np.random.default_rng(4711)
    pairs, ws = [], []
    for w in (1, 2, 7, 16, 33, 64, 128, 200, 256, 400, 512):
        for it in range(6):
            ql = int(rng.integers(w + 40, 900)) if it < 4 else int(rng.integers(2500, 7000))
            q = random_codes(rng, ql, 0.01 if it % 3 == 0 else 0.0)
            if it == 1:  # low complexity: many equal scores
                q = np.tile(random_codes(rng, 5), ql // 5 + 1)[:ql]
            t = mutate(rng, q, 0.05, 0.015, 0.015)
            cut = w + int(rng.integers(5, 300))
            if it % 2 == 0 and len(t) > cut + 20:  # target shorter: the band leaves through the target's end
                at = int(rng.integers(0, len(t) - cut))
                t = np.concatenate([t[:at], t[at + cut:]])
            else:  # target longer
                at = int(rng.integers(0, len(t)))
                t = np.concatenate([t[:at], random_codes(rng, cut), t[at:]])
            pairs.append((q, t))
            ws.append(w)
    res, cig = engine.align_pairs(pairs, w=ws, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    dropped = 0
    for (q, t), w, r in zip(pairs, ws, res):
        exp = oracle.extz2(q, t, w=w)
        got = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        for fld in ("score", "mte", "mte_q", "zdropped"):
            assert int(r[fld]) == exp[fld], (fld, w, len(q), len(t), int(r[fld]), exp[fld])
        if exp["zdropped"]:
            dropped += 1
            for fld in ("max", "max_t", "max_q"):
                assert int(r[fld]) == exp[fld], (fld, w, len(q), len(t), int(r[fld]), exp[fld])
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"]), (w, len(q), len(t))
    assert dropped >= 50


def test_generic_scoring_and_approximate_modes_fuzz(engine, oracle):
    """KSW_EZ_GENERIC_SC (scores from the whole matrix, also for the wildcard row), KSW_EZ_APPROX_MAX (one followed H
    value instead of H[]) and KSW_EZ_APPROX_DROP (z-drop on it): every ksw_extz_t field and the CIGAR against the
    oracle (pinned against the reference kernel for the same flags, tests/test_oracle_vs_ref.py)."""
    rng = np.random.default_rng(808)
    mat = np.array([6, -3, -5, -3, -1, -3, 6, -3, -5, -1, -5, -3, 6, -3, -1, -3, -5, -3, 6, -1, -1, -1, -1, -1, 1], np.int8)
    n = 0
    for flag in (0x04, 0x08, 0x18, 0x0c, 0x1c, 0x48, 0x09, 0x05):
        pairs, kws = [], []
        for _ in range(40):
            q = random_codes(rng, int(rng.integers(1, 500)), 0.03)
            d = rng.random() * 0.2
            t = mutate(rng, q, d, d / 3, d / 3)
            if rng.random() < 0.3:
                k = int(rng.integers(0, len(t)))
                t = np.concatenate([t[:k], random_codes(rng, int(rng.integers(1, 120))), t[k:]])
            pairs.append((q, t))
            kws.append(dict(w=int(rng.choice([-1, -1, 3, 17, 64])), zdrop=int(rng.choice([-1, 40, 300]))))
        m = mat if flag & 0x04 else sedef_mat()
        res, cig = engine.align_pairs(pairs, w=[k["w"] for k in kws], zdrop=[k["zdrop"] for k in kws], flag=flag, mat=m,
                                      gapo=12, gape=2)
        for (q, t), kw, r in zip(pairs, kws, res):
            exp = oracle.extz2(q, t, mat=m, gapo=12, gape=2, flag=flag, **kw)
            got = _rec_to_dict(r, cig)
            for f in FIELDS:
                assert got[f] == exp[f], (hex(flag), f, kw, len(q), len(t), got[f], exp[f])
            assert cigar_to_str(got["cigar"]) == cigar_to_str(exp["cigar"]), (hex(flag), kw, len(q), len(t))
            n += 1
    assert n == 320


def _full_size_properties(engine, batch, w, cpu_sample, oracle):
    """Size-independent checks on EVERY task of a batch (vectorised over the CIGAR pool) + an exact comparison with the
    CPU path on a sample: the CIGAR consumes both sequences, is a canonical run-length code, its counters add up, and --
    the sequences hold no N, full band -- the path it describes scores exactly the DP score: 5 per match, -4 per mismatch,
    -(40 + length) per gap (checked to hold for the oracle's own results)."""
    import sedef_amd
    from oracle.binding import Reference
    pool, q_off, qlen, t_off, tlen = batch
    n = len(qlen)
    tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
    tasks["w"], tasks["zdrop"] = w, -1
    res, cig = engine.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    nc = res["n_cigar"].astype(np.int64)
    off = res["cigar_off"].astype(np.int64)
    start = np.cumsum(nc) - nc
    ids = np.repeat(np.arange(n), nc)
    words = cig[np.repeat(off - start, nc) + np.arange(int(nc.sum()))]
    ops, lens = (words & 15).astype(np.int64), (words >> 4).astype(np.int64)
    assert ops.max() <= 2 and lens.min() >= 1
    same_task = ids[1:] == ids[:-1]
    assert not np.any(same_task & (ops[1:] == ops[:-1]))  # ksw_push_cigar merges equal neighbours (extern/ksw2.h:98-111)

    def per_task(x):
        return np.bincount(ids, weights=x, minlength=n).astype(np.int64)

    # a band that cannot reach the corner: no score, and the CIGAR -- if any -- ends at the best cell (extern/ksw2_extz2_sse.cc:286-296;
    # compared with the CPU path on the sample)
    done = (nc > 0) & (res["zdropped"] == 0) & (res["score"] > -0x40000000)
    assert np.array_equal(per_task(lens * (ops != 2))[done], np.asarray(qlen, np.int64)[done])
    assert np.array_equal(per_task(lens * (ops != 1))[done], np.asarray(tlen, np.int64)[done])
    assert np.array_equal(per_task(lens * (ops == 0)), res["matches"].astype(np.int64) + res["mismatches"])
    assert np.array_equal(per_task(lens * (ops != 0)), res["gap_bases"].astype(np.int64))
    assert np.array_equal(per_task((ops != 0).astype(np.int64)), res["gaps"].astype(np.int64))
    path_score = 5 * res["matches"].astype(np.int64) - 4 * res["mismatches"] - (40 * res["gaps"].astype(np.int64)
                                                                               + res["gap_bases"])
    # full band only: under a band the reference's score can exceed its own path's (cells outside the band keep stale
    # values, extern/ksw2_extz2_sse.cc:115-138; one such task in configs[4]'s 3,000, identical on the GPU)
    full = done & (np.broadcast_to(np.asarray(w, np.int32), (n,)) < 0)
    assert np.array_equal(path_score[full], res["score"].astype(np.int64)[full])
    try:
        cpu = Reference()
    except Exception:
        cpu = oracle
    ws = np.broadcast_to(np.asarray(w, np.int32), (n,))
    for wv in np.unique(ws[cpu_sample]):
        sel = cpu_sample[ws[cpu_sample] == wv]
        score, h = cpu.batch(pool, q_off[sel], qlen[sel], t_off[sel], tlen[sel], w=int(wv))
        assert np.array_equal(res["score"][sel], score), int(wv)
        for k, hk in zip(sel, h):
            assert _fnv(cig[off[k]:off[k] + nc[k]]) == int(hk), (int(k), int(qlen[k]), int(tlen[k]), int(wv))
    return res, done


def test_config4_full_size_one_million_tasks(engine, oracle):
    """configs[3] at the size bench.py --workload hg19mix runs: 1,000,000 tasks, the heavy ones up to 6000 x 6000."""
    import bench
    batch, w = bench.synth_hg19_mixture_fast(1000000, seed=7, big=6000)
    qlen = batch[1 + 1]
    rng = np.random.default_rng(1)
    heavy = np.nonzero(qlen >= 1200)[0]
    sample = np.unique(np.r_[rng.choice(len(qlen), 4000, replace=False), heavy[np.argsort(qlen[heavy])[:40]], heavy[-8:]])
    res, done = _full_size_properties(engine, batch, w, sample, oracle)
    assert done.all() and len(heavy) > 300 and len(qlen) == 1000000 and int(res["n_cigar"].astype(np.int64).sum()) > 2000000


def _device_bytes(eng, batch, w):
    """(result records, CIGAR words) of one batch call through the host-buffer entry point, as bytes."""
    import sedef_amd
    pool, q_off, qlen, t_off, tlen = batch
    tasks = np.zeros(len(qlen), sedef_amd.TASK_DTYPE)
    tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"], tasks["w"], tasks["zdrop"] = q_off, t_off, qlen, tlen, w, -1
    res, cig = eng.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
    return res.tobytes(), np.ascontiguousarray(cig).tobytes()


@pytest.mark.parametrize("switch", ["SDF_SPLIT_MIN", "SDF_EARLY_HEAVY", "SDF_NO_MIXED", "SDF_STRIP_COLS", "SDF_SCAN_POOL_FROM"])
def test_start_paths_on_and_off_give_the_same_bytes(engine, switch):
    """The paths that change HOW a large batch starts or pairs, never what it returns, each on against off on a batch that
    takes it: the two-part start of mid-size batches of one size (SDF_SPLIT_MIN=0: off), the early start of the heavy chunks
    of a large batch (SDF_EARLY_HEAVY=0: off), the mixed pairs (SDF_NO_MIXED=1: off), the chains' block width, the threaded scan
    of the cut.  Same records, same CIGAR words."""
    import bench
    if switch == "SDF_SPLIT_MIN":  # (the default rule wants the process's only context: asked for here)
        batch, w = bench.synth_batch(60000, 400, seed=9), 64
        engine, off = _engine_with_env(SDF_SPLIT_MIN=20000), _engine_with_env(SDF_SPLIT_MIN=0)
    elif switch == "SDF_EARLY_HEAVY":
        (batch, w) = bench.synth_hg19_mixture_fast(420000, seed=10, big=3000)
        off = _engine_with_env(SDF_EARLY_HEAVY=0)
    elif switch == "SDF_STRIP_COLS":  # (chains of few wavefronts take blocks of 256 columns by themselves: against 512)
        (batch, w) = bench.synth_hg19_mixture_fast(300000, seed=12, big=5000)
        off = _engine_with_env(SDF_STRIP_COLS=8)
    elif switch == "SDF_SCAN_POOL_FROM":  # (the threaded two-pass scan of batches from 120,000 tasks: against the one-thread scan)
        (batch, w) = bench.synth_hg19_mixture_fast(150000, seed=13, big=4000)
        off = _engine_with_env(SDF_SCAN_POOL_FROM=100000000)
    else:
        (batch, w) = bench.synth_mm8_mixture_fast(12000, seed=11, max_len=6000)
        off = _engine_with_env(SDF_NO_MIXED=1)
    a = _device_bytes(engine, batch, w)
    paired_on = engine.last_paired()
    b = _device_bytes(off, batch, w)
    assert a[0] == b[0] and a[1] == b[1]
    if switch == "SDF_NO_MIXED":
        assert paired_on > 8000 > off.last_paired()


def test_config5_full_size_mixed_bands(engine, oracle):
    """configs[4] at the size profiles/mix_probe.py mm8 runs: 3,000 tasks up to 20 kb, bands 64..512 in one batch."""
    import bench
    batch, w = bench.synth_mm8_mixture(3000, seed=8, max_len=20000)
    sample = np.arange(0, 3000, 10)
    res, done = _full_size_properties(engine, batch, w, sample, oracle)
    assert 2500 < int(done.sum()) < 3000  # the one-sided kb-scale indels leave their bands


def test_config5_throughput_batch_of_100000_tasks(oracle):
    """configs[4] at the size profiles/mix_probe.py mm8 100000 times (round 4: the batch whose banded tasks of different
    lengths run two to a wavefront): every task through the size-independent checks, a sample of 600 -- the longest tasks
    among them -- against the reference kernel.  The library's default workspace (`sdf_create(device, 0)`: half of the free HBM),
    as the probe: the batch's direction flags are 131 GB."""
    import bench
    import sedef_amd
    batch, w = bench.synth_mm8_mixture_fast(100000, seed=505)
    eng = sedef_amd.Extz2Engine(0)
    qlen = batch[2]
    rng = np.random.default_rng(3)
    sample = np.unique(np.r_[rng.choice(100000, 560, replace=False), np.argsort(qlen)[-40:]])
    res, done = _full_size_properties(eng, batch, w, sample, oracle)
    assert eng.last_paired() > 80000
    assert 90000 < int(done.sum()) < 100000  # (every 20th long task lost 1-5 kb of its target: its band runs out)


def test_config2_full_size_banded_and_full_band(engine, oracle):
    """configs[1] at the size bench.py times: the 100,000 seed-42 tasks at w = 128, and the same inputs in SEDEF's real
    mode (w = -1), every task through the size-independent checks, a sample against the CPU path."""
    import bench
    batch = bench.synth_batch(100000, 1000, seed=42)
    rng = np.random.default_rng(2)
    res, done = _full_size_properties(engine, batch, 128, np.sort(rng.choice(100000, 2000, replace=False)), oracle)
    assert done.all() and int(res["n_cigar"].astype(np.int64).sum()) > 3000000
    res, done = _full_size_properties(engine, batch, -1, np.sort(rng.choice(100000, 500, replace=False)), oracle)
    assert done.all()


def test_many_unpaired_banded_tasks_take_the_wave_kernel(engine, oracle):
    """More than 512 banded tasks of a chunk without a partner of their geometry are not paired with themselves: they fill
    launches of the one-task wave kernel (a wavefront whose halves compute the same task wastes half its row)."""
    rng = np.random.default_rng(4801)
    pairs, ws, seen = [], [], set()
    while len(pairs) < 640:
        ql = int(rng.integers(30, 900))
        tl = max(1, ql + int(rng.integers(-40, 40)))
        w = int(rng.choice([16, 33, 64, 100, 128, 200, 256]))
        if (ql, tl, w) in seen:
            continue
        seen.add((ql, tl, w))
        pairs += _same_geometry_tasks(rng, ql, tl, 1, 0.02)
        ws.append(w)
    _check_fast(engine, oracle, pairs, ws)
    assert engine.last_paired() == 0
    _check_fast(engine, oracle, pairs[:300], ws[:300])  # few of them: paired with themselves, as before


def _mixed_length_tasks(rng, n, ws, lo, hi, n_frac=0.01):
    """Banded tasks of the bands `ws` and lengths lo..hi (log-uniform), targets mutated copies with the odd indel or
    unrelated stretch -- and |qlen - tlen| small enough for most bands to reach the corner."""
    pairs, out_w = [], []
    for _ in range(n):
        w = int(rng.choice(ws))
        ql = int(np.exp(rng.uniform(np.log(lo), np.log(hi))))
        q = random_codes(rng, ql, n_frac if rng.random() < 0.3 else 0.0)
        d = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
        t = mutate(rng, q, d, d / 4, d / 4)
        if rng.random() < 0.3 and len(t) > 60:
            k, L = int(rng.integers(0, len(t) - 10)), int(rng.integers(1, max(2, w // 2)))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else np.concatenate([t[:k], t[k + L:]])
        if len(t) == 0:
            t = random_codes(rng, 1)
        pairs.append((q, t))
        out_w.append(w)
    return pairs, out_w


def test_mixed_pairs_banded_tasks_of_different_lengths(oracle):
    """Round 4: two banded tasks of one (w, flag) and DIFFERENT lengths per wavefront (extz2_pair.hip, MIXED): the rows on
    which neither task's sequence ends clip the band run side by side, each task's last ~w rows alone.  Every register
    count (2..9: windows of up to 576 slots, w = 512), sequences longer than the LDS windows, bands at the 16-cell
    block edges, N runs, indels at the band edge, tasks whose band runs out in the same batch (they keep their routes)."""
    eng = _engine_with_env(SDF_MIXED_MIN=2)
    rng = np.random.default_rng(4104)
    pairs, ws = _mixed_length_tasks(rng, 700, [15, 16, 17, 31, 33, 64, 100, 128, 129, 200, 256, 300, 448, 512], 40, 1500)
    p2, w2 = _mixed_length_tasks(rng, 60, [64, 128, 256, 512], 2000, 4200)  # (4000+ rows: back from the banded stripe kernel)
    pairs, ws = pairs + p2, ws + w2
    _check_fast(eng, oracle, pairs, ws)
    assert eng.last_paired() >= 500
    off = _engine_with_env(SDF_NO_MIXED=1)
    _check_fast(off, oracle, pairs[:200], ws[:200])
    assert off.last_paired() < eng.last_paired()
    # other scorings (the bytes wrap), score-only tasks next to CIGAR tasks
    sc = dict(mat=sedef_mat(2, -7), gapo=6, gape=3)
    _check_fast(eng, oracle, pairs[:300], ws[:300], **sc)
    _check_fast(eng, oracle, pairs[:300], ws[:300], mat=sedef_mat(11, -9), gapo=55, gape=4)
    flags = [int(rng.choice([0, 0, 1, 0x80])) for _ in range(300)]  # (score only; KSW_EZ_REV_CIGAR)
    _check_fast(eng, oracle, pairs[300:600], ws[300:600], flags=flags)


def test_brief_results_and_buffers_sized_once(oracle):
    """sdf_extz2_batch_brief returns (cigar_off, n_cigar, matches) of the full records; after sdf_reserve twenty batches within
    the bounds leave the context's device memory where the reserve put it (no growth after the first batch,
    nothing hoarded), and a batch beyond the bounds still runs (the buffers grow)."""
    import sedef_amd
    from sedef_amd.extz2 import TASK_DTYPE, RESERVE_BRIEF, RESERVE_ANCHORS
    eng = sedef_amd.Extz2Engine(0, workspace_bytes=2 << 30)
    eng.reserve(40000, 6_000_000, workspace_bytes=2 << 30, flags=RESERVE_BRIEF | RESERVE_ANCHORS)
    held = eng.device_bytes()
    assert held >= (2 << 30)
    rng = np.random.default_rng(4242)

    def batch(n, lo, hi):
        pairs = []
        for _ in range(n):
            q = random_codes(rng, int(rng.integers(lo, hi)), 0.002)
            pairs.append((q, mutate(rng, q, 0.05, 0.01, 0.01)))
        tasks = np.zeros(n, TASK_DTYPE)
        off, chunks = 0, []
        for k, (q, t) in enumerate(pairs):
            tasks["q_off"][k], tasks["qlen"][k] = off, len(q)
            off += len(q)
            tasks["t_off"][k], tasks["tlen"][k] = off, len(t)
            off += len(t)
            chunks += [q, t]
        tasks["w"], tasks["zdrop"] = -1, -1
        return pairs, tasks, np.concatenate(chunks)

    for it in range(20):
        pairs, tasks, pool = batch(int(rng.integers(200, 30000)) if it % 4 else 12000, 8, 120 if it % 3 else 400)
        brief, cig_b = eng.align_batch_brief(tasks, pool)
        if it % 5 == 0:
            full, cig_f = eng.align_batch(tasks, pool, want=sedef_amd.extz2.WANT_CIGAR | sedef_amd.extz2.WANT_SCORE)
            assert np.array_equal(cig_b, cig_f)
            for f in ("cigar_off", "n_cigar", "matches"):
                assert np.array_equal(brief[f], full[f].astype(brief[f].dtype)), f
            for k in range(0, len(pairs), 97):
                exp = oracle.extz2(pairs[k][0], pairs[k][1], w=-1)
                got = cig_b[int(brief["cigar_off"][k]):int(brief["cigar_off"][k]) + int(brief["n_cigar"][k])]
                assert cigar_to_str(got) == cigar_to_str(exp["cigar"])
        assert eng.device_bytes() == held, (it, eng.device_bytes(), held)
    # beyond the bounds: the buffers grow, the results stay right
    pairs, tasks, pool = batch(60000, 8, 200)
    brief, cig_b = eng.align_batch_brief(tasks, pool)
    for k in range(0, len(pairs), 1999):
        exp = oracle.extz2(pairs[k][0], pairs[k][1], w=-1)
        got = cig_b[int(brief["cigar_off"][k]):int(brief["cigar_off"][k]) + int(brief["n_cigar"][k])]
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"])
    assert eng.device_bytes() > held
    eng.close()


def _fasta_pool(rng, n_seq, lo, hi):
    """Raw FASTA characters: soft-masked ACGT, N runs and IUPAC letters (align_dna maps all of those to the wildcard)."""
    letters = np.frombuffer(b"ACGTacgtNnRYKMSWryBDHV-*", np.uint8)
    prob = np.array([20, 20, 20, 20, 4, 4, 4, 4, 1.0, 0.4] + [0.05] * 14)
    prob /= prob.sum()
    seqs = [letters[rng.choice(len(letters), int(rng.integers(lo, hi)), p=prob)] for _ in range(n_seq)]
    return seqs


def _align_codes(chars):
    """align_dna (reference: src/common.h:60-70,91) as the HOST applies it: table index c & 127."""
    tab = np.full(128, 4, np.uint8)
    for k, c in enumerate(b"ACGT"):
        tab[c] = tab[c | 0x20] = k
    return tab[chars & 127]


def test_resident_pool_tasks_equal_host_coded_tasks(oracle):
    """sdf_extz2_batch_pairs (round 6): tasks name BYTE RANGES of raw FASTA characters resident in HBM (what the anchors call
    uploaded); the device applies align_dna and packs them (seq_pack.hip).  Equal -- records and every CIGAR word -- to
    sdf_extz2_batch_brief / sdf_extz2_batch on the same ranges cut out and coded on the host (reference: src/align.cc:39-57,80-84),
    for overlapping ranges, every alignment of a range to the 32-base packing groups, soft-masked and IUPAC letters, a 60 kb
    chunk, pinned and pageable uploads; a sample against the oracle."""
    import sedef_amd
    from sedef_amd.extz2 import TASK_DTYPE, WANT_CIGAR, WANT_SCORE
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(6001)
    # pairs of related sequences (a mutated copy, characters kept) so that the DP has something to find
    pool_parts, spans = [], []
    off = 0
    for q in _fasta_pool(rng, 40, 300, 9000) + _fasta_pool(rng, 1, 70000, 70001):
        keep = rng.random(len(q)) > 0.03
        t = q[keep].copy()
        sub = rng.random(len(t)) < 0.05
        t[sub] = np.frombuffer(b"ACGTacgtN", np.uint8)[rng.integers(0, 9, int(sub.sum()))]
        spans.append((off, len(q), off + len(q), len(t)))
        off += len(q) + len(t)
        pool_parts += [q, t]
    chars = np.concatenate(pool_parts)
    tasks = []
    for (qo, ql, to, tl) in spans[:-1]:
        for _ in range(60):  # SEDEF's gap fills: short ranges at every offset, plus a few of hundreds of bases
            a = int(rng.integers(0, min(ql, tl) - 1))
            n1 = int(rng.integers(1, 70)) if rng.random() < 0.9 else int(rng.integers(70, 1200))
            n2 = max(1, n1 + int(rng.integers(-6, 7)))
            tasks.append((qo + a, min(n1, ql - a), to + min(a, tl - 1), min(n2, tl - min(a, tl - 1))))
    qo, ql, to, tl = spans[-1]
    tasks.append((qo, 60000, to, 60000))  # one chunk of Align::MAX_KSW_SEQ_LEN
    tasks.append((qo + 60000, ql - 60000, to + 60000, tl - 60000))
    t_res = np.zeros(len(tasks), TASK_DTYPE)
    t_host = np.zeros(len(tasks), TASK_DTYPE)
    code_parts, coff = [], 0
    for k, (a, n1, b, n2) in enumerate(tasks):
        t_res[k] = (a, b, n1, n2, -1, -1, 0, 0)
        t_host[k] = (coff, coff + n1, n1, n2, -1, -1, 0, 0)
        code_parts += [_align_codes(chars[a:a + n1]), _align_codes(chars[b:b + n2])]
        coff += n1 + n2
    codes_pool = np.concatenate(code_parts)
    brief_h, cig_h = eng.align_batch_brief(t_host, codes_pool)
    for pinned in (True, False):
        assert eng.pool_upload(chars.tobytes(), pinned=pinned) == len(chars)
        brief_r, cig_r = eng.align_batch_pairs(t_res)
        assert np.array_equal(cig_r, cig_h)
        for f in ("cigar_off", "n_cigar", "matches"):
            assert np.array_equal(brief_r[f], brief_h[f]), f
    full_r, cig_f = eng.align_batch_pairs(t_res, want=WANT_CIGAR | WANT_SCORE)
    full_h, cig_fh = eng.align_batch(t_host, codes_pool, want=WANT_CIGAR | WANT_SCORE)
    assert np.array_equal(cig_f, cig_fh)
    for f in full_r.dtype.names:
        assert np.array_equal(full_r[f], full_h[f]), f
    for k in range(0, len(tasks) - 2, 41):
        a, n1, b, n2 = tasks[k]
        exp = oracle.extz2(_align_codes(chars[a:a + n1]), _align_codes(chars[b:b + n2]), w=-1)
        got = cig_r[int(brief_r["cigar_off"][k]):int(brief_r["cigar_off"][k]) + int(brief_r["n_cigar"][k])]
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"]) and int(full_r["score"][k]) == exp["score"]
    # a range outside the resident pool is refused, an empty batch is not an error
    bad = t_res[:1].copy()
    bad["q_off"] = len(chars) - 3
    bad["qlen"] = 8
    with pytest.raises(sedef_amd.extz2.SdfError):
        eng.align_batch_pairs(bad)
    out, cig = eng.align_batch_pairs(t_res[:0])
    assert len(out) == 0 and len(cig) == 0
    eng.close()


def test_resident_pool_after_anchors_call(oracle):
    """sdf_anchors_batch leaves the characters it uploaded resident: the stage's order of calls -- anchors of a super-batch,
    then DP rounds on ranges of the same pool -- and the anchors of a pool that was uploaded first (seq_pool = NULL)."""
    import ctypes as C
    import sedef_amd
    from sedef_amd.extz2 import ANCHOR_DTYPE, ANCHOR_PAIR_DTYPE, TASK_DTYPE
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(6002)
    base = _fasta_pool(rng, 1, 5000, 5001)[0]
    q = base.copy()
    r = base[rng.random(len(base)) > 0.02].copy()
    pool = np.concatenate([q, r])
    desc = np.zeros(1, ANCHOR_PAIR_DTYPE)
    desc[0] = (0, len(q), len(q), len(r), 0, 0)

    def anchors(seq_pool):
        out = np.zeros(len(pool), ANCHOR_DTYPE)
        offs = np.zeros(2, np.int64)
        used = C.c_size_t(0)
        eng._check(eng.lib.sdf_anchors_batch(eng.ctx, desc.ctypes.data, 1, seq_pool, len(pool), 11, out.ctypes.data, len(out),
                                             offs.ctypes.data, C.byref(used)))
        return out[:used.value].copy()

    a_host = anchors(pool.tobytes())  # uploads, and leaves the pool resident
    assert len(a_host) > 5 and int(eng.lib.sdf_pool_bytes(eng.ctx)) == len(pool)
    t = np.zeros(3, TASK_DTYPE)
    t[0] = (10, len(q) + 10, 200, 190, -1, -1, 0, 0)
    t[1] = (0, len(q), len(q), len(r), -1, -1, 0, 0)
    t[2] = (4000, len(q) + 3900, 33, 35, -1, -1, 0, 0)
    brief, cig = eng.align_batch_pairs(t)
    for k in range(3):
        exp = oracle.extz2(_align_codes(pool[t["q_off"][k]:t["q_off"][k] + t["qlen"][k]]),
                           _align_codes(pool[t["t_off"][k]:t["t_off"][k] + t["tlen"][k]]), w=-1)
        got = cig[int(brief["cigar_off"][k]):int(brief["cigar_off"][k]) + int(brief["n_cigar"][k])]
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"])
    eng.pool_upload(pool.tobytes())
    a_res = anchors(None)  # the pairs' offsets point into the resident pool
    assert np.array_equal(a_res, a_host)
    eng.close()


def test_view_entries_and_anchors_in_two_halves(oracle):
    """The entry points the stage driver binds since round 6 (host/pipeline.cc: GpuProvider): sdf_anchors_batch_view on the
    first half of a super-batch's pairs, sdf_anchors_batch_more on the second half BEHIND the first half's anchors (which must
    stay what they were), sdf_extz2_batch_pairs_view -- against the copying forms of the same calls on the same resident pool."""
    import ctypes as C
    import sedef_amd
    from sedef_amd.extz2 import ANCHOR_DTYPE, ANCHOR_PAIR_DTYPE, BRIEF_DTYPE, TASK_DTYPE, _scoring, sedef_mat
    eng = sedef_amd.Extz2Engine(0)
    lib = eng.lib
    lib.sdf_anchors_batch_view.restype = C.c_int
    lib.sdf_anchors_batch_view.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                           C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_size_t)]
    lib.sdf_anchors_batch_more.restype = C.c_int
    lib.sdf_anchors_batch_more.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t,
                                           C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_size_t)]
    lib.sdf_extz2_batch_pairs_view.restype = C.c_int
    lib.sdf_extz2_batch_pairs_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    rng = np.random.default_rng(6003)
    n = 300
    parts, off = [], 0
    desc = np.zeros(n, ANCHOR_PAIR_DTYPE)
    for k in range(n):
        m = int(np.exp(rng.uniform(np.log(40), np.log(9000))))
        q = _fasta_pool(rng, 1, m, m + 1)[0]
        r = q[rng.random(len(q)) > 0.03].copy()
        sub = rng.random(len(r)) < 0.04
        r[sub] = np.frombuffer(b"ACGTacgtN", np.uint8)[rng.integers(0, 9, int(sub.sum()))]
        same = int(k % 3 == 0)
        desc[k] = (off, off + len(q), len(q), len(r), same, int(rng.integers(-50, 50)) if same else 0)
        off += len(q) + len(r)
        parts += [q, r]
    pool = np.concatenate(parts)

    # the copying form, one call over all pairs (uploads the pool and leaves it resident)
    out_all = np.zeros(len(pool), ANCHOR_DTYPE)
    off_all = np.zeros(n + 1, np.int64)
    used = C.c_size_t(0)
    eng._check(lib.sdf_anchors_batch(eng.ctx, desc.ctypes.data, n, pool.tobytes(), len(pool), 11, out_all.ctypes.data,
                                     len(out_all), off_all.ctypes.data, C.byref(used)))
    out_all = out_all[:used.value]
    assert used.value == off_all[n] and used.value > n

    def view(ptr, count):
        buf = (C.c_char * (count * ANCHOR_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, ANCHOR_DTYPE, count)

    # the stage's form: pinned pool filled in place, first half as a view, second half behind it
    eng.pool_upload(pool.tobytes())
    h = n // 2
    p1, p2 = C.c_void_p(), C.c_void_p()
    off1, off2 = np.zeros(h + 1, np.int64), np.zeros(n - h + 1, np.int64)
    u1, u2 = C.c_size_t(0), C.c_size_t(0)
    eng._check(lib.sdf_anchors_batch_view(eng.ctx, desc[:h].ctypes.data, h, None, len(pool), 11, C.byref(p1), off1.ctypes.data,
                                          C.byref(u1)))
    first = view(p1, u1.value).copy()
    assert np.array_equal(off1, off_all[:h + 1]) and np.array_equal(first, out_all[:u1.value])
    eng._check(lib.sdf_anchors_batch_more(eng.ctx, desc[h:].ctypes.data, n - h, len(pool), 11, u1.value, C.byref(p2),
                                          off2.ctypes.data, C.byref(u2)))
    assert p2.value == p1.value + u1.value * ANCHOR_DTYPE.itemsize  # the same staging, behind the first half's anchors
    both = view(p1, u1.value + u2.value)
    assert np.array_equal(both[:u1.value], first), "the first half's anchors changed under the second call"
    assert np.array_equal(off2, off_all[h:] - off_all[h])  # (counted from *out)
    assert np.array_equal(both, out_all)

    # DP on ranges of the same pool: the view form against the copying form
    t = np.zeros(4000, TASK_DTYPE)
    pk = rng.integers(0, n, len(t))
    a = (rng.random(len(t)) * np.maximum(1, np.minimum(desc["qlen"][pk], desc["rlen"][pk]) - 2)).astype(np.int64)
    n1 = np.minimum(rng.integers(1, 90, len(t)), desc["qlen"][pk] - a)
    n2 = np.minimum(np.maximum(1, n1 + rng.integers(-5, 6, len(t))), desc["rlen"][pk] - a)
    t["q_off"], t["t_off"], t["qlen"], t["tlen"] = desc["q_off"][pk] + a, desc["r_off"][pk] + a, n1, n2
    t["w"], t["zdrop"] = -1, -1
    brief, cig = eng.align_batch_pairs(t)
    sc = _scoring(sedef_mat(), 40, 1)
    pb, pc, cu = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
    eng._check(lib.sdf_extz2_batch_pairs_view(eng.ctx, C.byref(sc), t.ctypes.data, len(t), C.byref(pb), C.byref(pc), C.byref(cu)))
    vb = np.frombuffer((C.c_char * (len(t) * 16)).from_address(pb.value), BRIEF_DTYPE, len(t))
    vc = np.frombuffer((C.c_char * (cu.value * 4)).from_address(pc.value), np.uint32, cu.value)
    assert cu.value == len(cig) and np.array_equal(vc, cig)
    for f in BRIEF_DTYPE.names:
        assert np.array_equal(vb[f], brief[f]), f
    for k in range(0, len(t), 397):
        exp = oracle.extz2(_align_codes(pool[t["q_off"][k]:t["q_off"][k] + t["qlen"][k]]),
                           _align_codes(pool[t["t_off"][k]:t["t_off"][k] + t["tlen"][k]]), w=-1)
        got = vc[int(vb["cigar_off"][k]):int(vb["cigar_off"][k]) + int(vb["n_cigar"][k])]
        assert cigar_to_str(got) == cigar_to_str(exp["cigar"])
    # `keep` beyond the staging: refused
    rc = lib.sdf_anchors_batch_more(eng.ctx, desc[h:].ctypes.data, n - h, len(pool), 11, 1 << 40, C.byref(p2),
                                    off2.ctypes.data, C.byref(u2))
    assert rc < 0 and not p2.value
    eng.close()
