"""Scope row f4: per-alignment columns of `stats generate` (reference: src/stats_main.cc:228-270 over the column
strings of populate_nice_alignment, src/align.cc:274-315).

CPU: oracle/stats_oracle.c against the golden vectors the reference's Alignment(fa, fb, cigar) produced (column
strings + AlignmentError counters: pinned), live against that constructor when the reference is present, and its column
loop against an independent numpy statement of the same counters.  GPU: sdf_stats_columns_batch through the C ABI
against the oracle, bit-exact."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle.binding import STATS_FIELDS
from util import random_stats_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _words(runs):
    return np.array([(l << 4) | op for op, l in runs], np.uint32)


def _parse(cig):
    runs, num = [], 0
    for ch in cig:
        if ch.isdigit():
            num = 10 * num + int(ch)
        else:
            runs.append(("MDI".index(ch), num))
            num = 0
    return runs


@pytest.fixture(scope="module")
def stats_golden():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "stats_columns_kat.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def _numpy_columns(aa, ab):
    """The counters of src/stats_main.cc:228-270 from the two column strings, vectorised (independent of stats_oracle.c)."""
    x = np.frombuffer(aa.encode("latin1"), np.uint8).astype(np.int32)
    y = np.frombuffer(ab.encode("latin1"), np.uint8).astype(np.int32)
    isup_x, isup_y = (x >= 65) & (x <= 90), (y >= 65) & (y <= 90)
    ux = np.where((x >= 97) & (x <= 122), x - 32, x)
    uy = np.where((y >= 97) & (y <= 122), y - 32, y)
    dash, N = ord("-"), ord("N")
    both = (ux != dash) & (uy != dash)
    mis = both & (ux != uy)
    pur_x = (ux == ord("A")) | (ux == ord("G"))
    ts = mis & np.where(pur_x, (uy == ord("A")) | (uy == ord("G")), (uy == ord("C")) | (uy == ord("T")))
    ceq = both & (ux != N) & (uy != N) & (ux == uy)
    return dict(indel_a=int((ux == dash).sum()), indel_b=int((uy == dash).sum()), aln_b=int(both.sum()),
                match_b=int(((ux != dash) & (ux == uy)).sum()), mismatch_b=int(mis.sum()), transitions_b=int(ts.sum()),
                transversions_b=int((mis & ~ts).sum()), uppercase_a=int(((x != dash) & (ux != N) & isup_x).sum()),
                uppercase_b=int(((y != dash) & (uy != N) & isup_y).sum()),
                uppercase_matches=int((both & (ux == uy) & isup_x & isup_y).sum()), matches=int(ceq.sum()),
                mismatches=int((both & ~ceq).sum()), span=len(x))


def test_oracle_equals_reference_golden(oracle, stats_golden):
    assert len(stats_golden) >= 300
    kinds = set()
    for c in stats_golden:
        runs = _parse(c["cigar"])
        out, aa, ab = oracle.stats_columns(c["a"], c["b"], _words(runs), want_columns=True)
        got = dict(zip(STATS_FIELDS, out.tolist()))
        assert (aa, ab) == (c["align_a"], c["align_b"])
        assert [got["matches"], got["mismatches"], got["gaps"], got["gap_bases"], got["span"]] == c["counts"]
        assert got["flags"] == 0
        exp = _numpy_columns(c["align_a"], c["align_b"])
        assert {k: got[k] for k in exp} == exp
        kinds.add((len(runs) > 64, any(l == 0 for _, l in runs), any(l > 64 for _, l in runs), len(runs) == 0))
    assert len(kinds) >= 6


def test_oracle_equals_reference_live(oracle):
    from oracle.binding import ReferenceAlign
    try:
        ref = ReferenceAlign()
    except (FileNotFoundError, OSError):
        pytest.skip("reference checkout absent (GPU box): golden vectors cover this")
    rng = np.random.default_rng(77)
    for k in range(300):
        a, b, runs = random_stats_case(rng, k)
        cig = "".join("%d%s" % (l, "MDI"[op]) for op, l in runs)
        raa, rab, cnt = ref.alignment_from_cigar(a, b, cig)
        out, aa, ab = oracle.stats_columns(a, b, _words(runs), want_columns=True)
        assert (aa, ab) == (raa, rab)
        assert out[10:15].tolist() == cnt


def test_oracle_refuses_a_cigar_longer_than_the_sequences(oracle):
    out = oracle.stats_columns("ACGT", "ACGT", _words([(0, 5)]))
    assert out[15] == 1
    out = oracle.stats_columns("ACGT", "ACG", _words([(0, 3), (2, 1)]))
    assert out[15] == 1
    out = oracle.stats_columns("ACGT", "ACG", _words([(0, 3), (1, 1)]))
    assert out[15] == 0 and out[14] == 4


def _check_batch(eng, oracle, cases):
    got = eng.stats_columns_batch([(a, b, _words(runs)) for a, b, runs in cases])
    for k, (a, b, runs) in enumerate(cases):
        exp = oracle.stats_columns(a, b, _words(runs))
        assert [int(got[k][f]) for f in STATS_FIELDS] == exp.tolist(), (k, len(a), len(b), len(runs))


@pytest.mark.gpu
def test_gpu_stats_columns_equal_oracle_on_golden(oracle, stats_golden):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    _check_batch(eng, oracle, [(c["a"], c["b"], _parse(c["cigar"])) for c in stats_golden])
    # the reference's own counters once more, straight from the fixture
    got = eng.stats_columns_batch([(c["a"], c["b"], _words(_parse(c["cigar"]))) for c in stats_golden])
    for g, c in zip(got, stats_golden):
        assert [int(g[f]) for f in ("matches", "mismatches", "gaps", "gap_bases", "span")] == c["counts"]


@pytest.mark.gpu
def test_gpu_stats_columns_fuzz_and_edges(oracle):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(5)
    cases = [random_stats_case(rng, k) for k in range(1200)]
    # edges: nothing at all, empty CIGAR over sequences, only gaps, one column, zero-length runs only, all N, exactly
    # 64 / 128 runs, one run of 100,000 columns, lower case only, bytes outside the alphabet
    cases += [("", "", []), ("ACGT", "ACG", []), ("ACGT", "ACG", [(1, 4), (2, 3)]), ("a", "A", [(0, 1)]),
              ("AC", "AC", [(1, 0), (2, 0), (0, 0)]), ("NNNN", "nnnn", [(0, 4)]),
              ("ACGT" * 16, "ACGA" * 16, [(0, 1)] * 64), ("ACGT" * 32, "ACGA" * 32, [(0, 1)] * 128),
              ("ACGT" * 25000, "ACGT" * 25000, [(0, 100000)]), ("acgt" * 50, "acga" * 50, [(0, 200)]),
              ("AC-T*R", "AC-T*Y", [(0, 6)]),
              # bytes outside ASCII (the per-column path of the kernel), next to units that take the packed path
              (b"AC\xc3\x80GTacgtNNAC\xffTACGTACGTAC", b"AC\xc3\x81GTacgaNnAC\xffTACGAACGTAC", [(0, 12), (1, 2), (0, 12)]),
              (bytes(range(1, 256)), bytes(range(1, 256)), [(0, 255)]),
              (bytes(range(1, 256)), bytes(reversed(range(1, 256))), [(0, 100), (2, 55), (1, 55), (0, 100)]),
              (bytes(range(33, 128)) * 3, (bytes(range(33, 128)) * 3).swapcase(), [(0, 285)]),
              ("@[`{AZaz" * 9, "@[`{azAZ" * 9, [(0, 72)])]
    _check_batch(eng, oracle, cases)
    assert len(eng.stats_columns_batch([])) == 0


@pytest.mark.gpu
def test_gpu_stats_columns_long_alignment_and_many(oracle):
    """One 3 Mb alignment with 200,000 runs, and 50,000 short alignments in one call."""
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(6)
    runs = []
    for _ in range(100000):
        runs += [(0, int(rng.integers(1, 40))), (int(rng.integers(1, 3)), int(rng.integers(1, 6)))]
    na = sum(l for op, l in runs if op != 2)
    nb = sum(l for op, l in runs if op != 1)
    a = rng.choice(list(b"ACGTacgtNn"), na).astype(np.uint8).tobytes().decode()
    b = rng.choice(list(b"ACGTacgtNn"), nb).astype(np.uint8).tobytes().decode()
    _check_batch(eng, oracle, [(a, b, runs)])
    small = [random_stats_case(rng, 8 * k + 2) for k in range(2000)] * 25
    got = eng.stats_columns_batch([(x, y, _words(r)) for x, y, r in small])
    for k in range(2000):
        exp = oracle.stats_columns(small[k][0], small[k][1], _words(small[k][2]))
        for rep in range(0, 25, 6):
            assert [int(got[rep * 2000 + k][f]) for f in STATS_FIELDS] == exp.tolist()


@pytest.mark.gpu
def test_gpu_stats_columns_long_and_short_in_one_call(oracle):
    """Alignments of more than 1,024 runs are cut into pieces of 512 runs by the host-buffer call: next to short ones, with
    zero-length runs at the cuts, a CIGAR that stops short of the sequences, and gap-only pieces."""
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(8)
    cases = []
    for n_runs in (1024, 1025, 1536, 1537, 5000, 3000):
        runs = []
        for k in range(n_runs):
            op = int(rng.choice([0, 0, 1, 2]))
            runs.append((op, 0 if k % 512 in (0, 511) and rng.random() < 0.5 else int(rng.integers(0, 30))))
        if n_runs == 3000:  # 600 gap runs in a row: a piece that consumes nothing of b, then one that consumes nothing of a
            runs[1000:1300] = [(1, 3)] * 300
            runs[1300:1600] = [(2, 2)] * 300
        na = sum(l for op, l in runs if op != 2) + 17
        nb = sum(l for op, l in runs if op != 1) + 5
        a = rng.choice(list(b"ACGTacgtNn-"), na).astype(np.uint8).tobytes()
        b = rng.choice(list(b"ACGTacgtNn"), nb).astype(np.uint8).tobytes()
        cases.append((a, b, runs))
        cases.append(random_stats_case(rng, n_runs))
    _check_batch(eng, oracle, cases)
    a, b, runs = cases[8]
    with pytest.raises(sedef_amd.SdfError, match="alignment 1: the CIGAR does not fit"):
        eng.stats_columns_batch([("ACGT", "ACGT", _words([(0, 4)])), (a, b[:len(b) - 200], _words(runs))])
    with pytest.raises(sedef_amd.SdfError, match="alignment 0: the CIGAR does not fit"):
        eng.stats_columns_batch([(a, b, np.concatenate([_words(runs[:2000]), np.array([(1 << 4) | 7], np.uint32)]))])


@pytest.mark.gpu
def test_gpu_stats_columns_segment_list_overflow(oracle):
    """The device cuts long alignments into segments that a second launch counts; with a list of four segments most long
    alignments of a call find it full and are counted by their own wavefront: same counters either way."""
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0, config=dict(SDF_STATS_ITEMS=4))
    rng = np.random.default_rng(9)
    cases = [random_stats_case(rng, n_runs) for n_runs in (1100, 1500, 30, 2100, 1025, 700, 4000)]
    _check_batch(eng, oracle, cases)


@pytest.mark.gpu
def test_gpu_stats_columns_errors():
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    with pytest.raises(sedef_amd.SdfError, match="does not fit"):
        eng.stats_columns_batch([("ACGT", "ACGT", _words([(0, 4)])), ("ACGT", "ACGT", _words([(0, 5)]))])
    with pytest.raises(sedef_amd.SdfError, match="does not fit"):
        eng.stats_columns_batch([("ACGT", "ACGT", np.array([(4 << 4) | 3], np.uint32))])
    # the engine is usable afterwards
    out = eng.stats_columns_batch([("ACGT", "ACGA", _words([(0, 4)]))])
    assert int(out[0]["match_b"]) == 3 and int(out[0]["transitions_b"]) == 0 and int(out[0]["transversions_b"]) == 1
