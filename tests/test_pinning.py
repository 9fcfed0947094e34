"""Pins for the host rows that do compile from the reference (FASTA access, Hit::extend, Sequence) and independent
brute-force checks of seed anchors and chaining (tests/bruteforce.py) for both the host code and the HIP kernels.

generate_anchors / chain_anchors of the reference (src/chain.cc) cannot be compiled here (Boost via src/search.h): the
brute-force checkers are written from the definition of their result instead of from either implementation."""
import gzip
import json
import os

import numpy as np
import pytest

import bruteforce
import hostgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from sedef_amd import host as h
    from sedef_amd.build import build_library
    build_library()
    h.build_host()
    return h


@pytest.fixture(scope="module")
def host_golden():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "host_align_kat.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def _write_fasta(tmp_path, case, k):
    fa = tmp_path / ("g%d.fa" % k)
    fa.write_text(case["text"])
    with open(str(fa) + ".fai", "w") as f:
        for (name, length, offset, lb, ll) in case["entries"]:
            f.write("%s some description\t%d\t%d\t%d\t%d\n" % (name, length, offset, lb, ll))
    return str(fa)


# ---------------------------------------------------------------- D3 / D2: reference-generated golden vectors
def test_fasta_get_sequence_matches_reference_golden(host, host_golden, tmp_path):
    total = 0
    for k, case in enumerate(host_golden["fastas"]):
        fa = _write_fasta(tmp_path, case, k)
        for qy in case["queries"]:
            seq, end = host.fasta_get(fa, qy["name"], qy["start"], qy["end"])
            assert seq == qy["seq"] and end == qy["end_out"], (k, qy["name"], qy["start"], qy["end"])
            total += 1
    assert total >= 200


def test_hit_extend_matches_reference_golden(host, host_golden):
    assert len(host_golden["extends"]) >= 200
    for c in host_golden["extends"]:
        assert host.hit_extend(*c["io"], c["factor"], c["max_extend"]) == c["expect"]


def test_sequence_ctor_matches_reference_golden(host, host_golden):
    for c in host_golden["sequences"]:
        assert list(host.sequence(c["name"], c["seq"])) == c["expect"]
    # is_rc = true reverse-complements (src/hash.cc:106-108 -> rc(), src/util.cc:43-48: case kept, non-ACGT -> N)
    assert host.sequence("x", "AACgtNry", True) == ("x", "NNNacGTT", True)


def test_fasta_and_extend_live_vs_reference(host, tmp_path):
    from oracle.binding import ReferenceAlign
    try:
        ref = ReferenceAlign()
    except Exception:
        pytest.skip("oracle/_ref/libref_align.so not built and /root/reference absent")
    rng = np.random.default_rng(99)
    for k, (lb, nchr) in enumerate(((61, 2), (13, 4), (80, 1))):
        text, entries = hostgen.fasta_text(rng, nchr, lb)
        fa = _write_fasta(tmp_path, dict(text=text, entries=entries), k)
        bare = tmp_path / ("bare%d.fa" % k)  # the reference driver needs a FASTA without .fai (see ref_fasta_get)
        bare.write_text(text)
        for (name, length, offset, lb_, ll) in entries:
            for (a, b) in hostgen.fasta_queries(rng, length, lb_):
                assert host.fasta_get(fa, name, a, b) == ref.fasta_get(str(bare), name, length, offset, lb_, ll, a, b)
    for _ in range(300):
        io = [int(x) for x in rng.integers(0, 50000, 4)]
        io[1] += io[0] + 1
        io[3] += io[2] + 1
        f, mx = float(rng.uniform(0.1, 6)), int(rng.integers(1, 20000))
        assert host.hit_extend(*io, f, mx) == ref.hit_extend(*io, f, mx)


# ---------------------------------------------------------------- C1: anchors vs. brute force
def _anchor_cases(rng, n):
    cases = []
    for it in range(n):
        q = hostgen.rseq(rng, int(rng.integers(5, 1500)), 0.004 if it % 2 else 0.0)
        r = hostgen.rseq(rng, int(rng.integers(0, 200))) + hostgen.mut(rng, q, rng.random() * 0.1) + \
            hostgen.rseq(rng, int(rng.integers(0, 200)))
        if it % 5 == 0:  # k-mers with >= 1000 reference occurrences are skipped (src/chain.cc:61)
            rep = "ACGTTGCAACGT" * 100
            q, r = q[:300] + rep[:700] + q[300:], r[:200] + rep + rep[:500] + r[200:]
        if it % 7 == 0:  # a soft-masked stretch: has_u == 0 anchors
            q, r = q.lower(), r[:len(r) // 2].lower() + r[len(r) // 2:]
        same = it % 3 == 0
        cases.append((q, r, same, int(rng.integers(-30, 30)) if same else 0))
    cases.append(("ACGT", "ACGTACGTACGTACGT", False, 0))
    cases.append(("ACGTACGTAGCTAGCTAGCATCGAT", "ACGTACGTAGCTAGCTAGCATCGAT", True, 0))  # whole main diagonal skipped
    cases.append(("ACGTACGTAGCTAGCTAGCATCGATNNACGATCGATCAGCTACGACTAGCAT", "ACGTACGTAGCTAGCTAGCATCGATNNACGATCGATCAGCTACGACTAGCAT",
                  False, 0))
    return cases


def test_host_anchors_equal_bruteforce(host):
    rng = np.random.default_rng(31)
    tot = 0
    for (q, r, same, delta) in _anchor_cases(rng, 24):
        for k in (11, 8):
            exp = bruteforce.anchors_bruteforce(q, r, k, same, 0, delta)
            got = host.anchors(q, r, k, same_chr=same, qstart=0, rstart=delta)
            assert got == exp, (len(q), len(r), same, delta, k, len(got), len(exp))
            tot += len(exp)
    assert tot > 2000


@pytest.mark.gpu
def test_gpu_anchors_equal_bruteforce(host):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(32)
    cases = _anchor_cases(rng, 40)
    for k in (11, 9):
        got = eng.anchors_batch(cases, k)
        for (q, r, same, delta), g in zip(cases, got):
            assert g == bruteforce.anchors_bruteforce(q, r, k, same, 0, delta), (len(q), len(r), same, delta, k)


def _np_seq(rng, n):
    return np.frombuffer(b"ACGTacgt", np.uint8)[rng.integers(0, 8, n)].tobytes().decode()


@pytest.mark.gpu
def test_gpu_anchors_k_up_to_15_long_sequences_and_many_pairs(host):
    """Round 4: the cliffs of sdf_anchors_batch are gone -- k up to 15 (`-k` is a CLI parameter of the reference,
    src/align_main.cc:366-370), references beyond 4 Mb (the position field of the sort key takes the bits the call's longest
    reference needs), more than 65,535 pairs a call (pairs beyond the bits left of hash and position run range by range)."""
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(33)
    cases = _anchor_cases(rng, 16)
    for k in (8, 12, 14, 15):
        got = eng.anchors_batch(cases, k)
        for (q, r, same, delta), g in zip(cases, got):
            assert g == bruteforce.anchors_bruteforce(q, r, k, same, 0, delta), (len(q), len(r), same, delta, k)
    # a 5.2 Mb reference (23 position bits: at k = 15 eleven bits are left for the pair index -> 2,048 pairs a range) with
    # 2,300 small pairs behind it: three ranges; expected = the host's sort-merge join (equal to the brute force above)
    r_big = _np_seq(rng, 5200000)
    q_big = "".join(hostgen.mut(rng, r_big[a:a + 2000], 0.03) + _np_seq(rng, 500)
                    for a in rng.integers(0, 5000000, 60).tolist())
    small = [(hostgen.rseq(rng, int(rng.integers(20, 400))),) for _ in range(2300)]
    small = [(q[0], hostgen.mut(rng, q[0], 0.05) + hostgen.rseq(rng, 30), False, 0) for q in small]
    cases = [(q_big, r_big, False, 0)] + small
    got = eng.anchors_batch(cases, 15)
    assert len(got[0]) > 2000
    for idx in [0] + rng.choice(np.arange(1, len(cases)), 150, replace=False).tolist() + [len(cases) - 1, 2047, 2048, 2049]:
        q, r, same, delta = cases[idx]
        assert got[idx] == host.anchors(q, r, 15, same_chr=same, qstart=0, rstart=delta), idx
    # 70,000 pairs in one call (gridDim.y holds 65,535)
    tiny = []
    for _ in range(70000):
        q = _np_seq(rng, int(rng.integers(12, 60)))
        tiny.append((q, q[3:] + "ACGT", False, 0))
    got = eng.anchors_batch(tiny, 8)
    for idx in rng.choice(70000, 300, replace=False).tolist() + [0, 65534, 65535, 65536, 69999]:
        q, r, same, delta = tiny[idx]
        assert got[idx] == bruteforce.anchors_bruteforce(q, r, 8, same, 0, delta), idx


# ---------------------------------------------------------------- C2: chaining vs. brute force
def _chain_cases(host, rng, n_real, n_rand):
    cases = [np.zeros((0, 4), np.int32), np.array([[5, 9, 11, 1]], np.int32),
             np.array([[0, 0, 12, 0], [30, 31, 15, 1]], np.int32),
             np.array([[0, 0, 12, 0], [30, 31, 15, 1], [400, 80, 11, 0]], np.int32)]
    for it in range(n_real):
        q = hostgen.rseq(rng, int(rng.integers(50, 5000)), 0.004 if it % 2 else 0.0)
        r = hostgen.rseq(rng, int(rng.integers(0, 300))) + hostgen.mut(rng, q, rng.random() * 0.15) + \
            hostgen.rseq(rng, int(rng.integers(0, 300)))
        if it % 4 == 0:
            rep = "ACGTTGCAACGT" * 30
            q, r = q[:200] + rep + q[200:], r[:100] + rep + r[100:]
        cases.append(np.array(host.anchors(q, r, 11), np.int32).reshape(-1, 4))
    for it in range(n_rand):  # random anchor sets: many equal scores and coordinates
        m = int(rng.integers(1, 300))
        span = int(rng.integers(20, 3000))
        cases.append(np.stack([rng.integers(0, span, m), rng.integers(0, span, m), rng.integers(11, 14, m),
                               rng.integers(0, 2, m)], 1).astype(np.int32))
    return cases


def test_host_chains_pass_bruteforce_check(host, oracle):
    rng = np.random.default_rng(41)
    unique = 0
    cases = _chain_cases(host, rng, 24, 24)
    for gap, score in ((210, 4), (50, 3)):
        for a in cases:
            chk = bruteforce.ChainCheck(a, gap, score)
            path, bounds = host.chain_raw(a, gap, score)
            chk.check(path, bounds)
            # ... and, ties included, exactly the chains of the oracle, whose tree is pinned on the reference's class
            # (tests/test_chain_oracle.py)
            chk.check_exact(path, bounds, oracle.chain_anchors(a, gap, score))
            exp = chk.expected_if_unique()
            if exp is not None:
                unique += 1
                assert list(path) == exp[0] and [tuple(b) for b in bounds] == exp[1]
    assert unique >= 20  # the exact comparison did run on tie-free inputs


@pytest.mark.gpu
def test_gpu_chains_pass_bruteforce_check(host, oracle):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(42)
    cases = _chain_cases(host, rng, 20, 20)
    for gap, score in ((210, 4), (1000, 5)):
        got = eng.chain_batch(cases, gap, score)
        for a, (gp, gb) in zip(cases, got):
            chk = bruteforce.ChainCheck(a, gap, score)
            chk.check(gp, gb)
            chk.check_exact(gp, gb, oracle.chain_anchors(a, gap, score))
            exp = chk.expected_if_unique()
            if exp is not None:
                assert list(gp) == exp[0] and [tuple(b) for b in gb] == exp[1]
