"""C2 / f3: the tie rules of anchor chaining, pinned on the reference's own SegmentTree class.

Which of several equally good predecessors chain_anchors gives an anchor is decided by the priority search tree of
src/segment.tpp (the `>=` / `>` at :62, :89, :128): by its shape and its history, not by a closed rule.  The reference
class compiles here without Boost (oracle/ref_align_driver.cc: ref_segtree_script); three restatements are compared with
it on scripts of activate / deactivate / rmq calls, for EQUALITY of every returned point and score and of every node's
pointer after the script:
  * the oracle's tree (oracle/chain_oracle.c)                      -- CPU, golden + live
  * the host's RangeMax (sedef_amd/csrc/host/chain.cc)             -- CPU, golden + live
  * the device tree of chain.hip (sdf_debug_chain_tree_script)     -- GPU, golden
and chain_anchors itself (host: sdfh_chain_raw; GPU: sdf_chain_batch) must return the oracle's path and boundaries
exactly, ties included; the oracle's sweep is checked against the O(n^2) definition (tests/bruteforce.py)."""
import gzip
import json
import os

import numpy as np
import pytest

import bruteforce
import hostgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kat():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "segtree_kat.json.gz"), "rb") as f:
        return json.loads(f.read().decode())["cases"]


@pytest.fixture(scope="module")
def host():
    from sedef_amd import host as h
    h.load_host()
    return h


@pytest.fixture(scope="module")
def ref_tree():
    from oracle.binding import ReferenceAlign
    try:
        return ReferenceAlign()
    except (FileNotFoundError, OSError):
        pytest.skip("reference checkout / oracle/_ref/libref_align.so not available")


def _same(got, case):
    out, state = got
    return np.array_equal(out, np.array(case["out"], np.int32).reshape(-1, 2)) and \
        np.array_equal(state, np.array(case["state"], np.int32))


def test_fixture_has_ties(kat):
    assert len(kat) >= 90 and sum(c["kind"] == "sweep" for c in kat) >= 25
    assert sum(c.get("hits", 0) for c in kat) >= 1000  # range queries of the sweeps that found a predecessor


def test_oracle_tree_equals_reference_golden(oracle, kat):
    for c in kat:
        assert _same(oracle.segtree_script(c["pts"], c["ops"]), c), c["kind"]


def test_host_tree_equals_reference_golden(host, kat):
    for c in kat:
        assert _same(host.rangemax_script(c["pts"], c["ops"]), c), c["kind"]


def test_sweep_scripts_are_the_oracles_call_streams(oracle, kat):
    """The stored `sweep` scripts are what the oracle's chain_anchors asks the tree (and the answers, the reference
    tree's, are what it got): its predecessors are the reference tree's choices."""
    for c in kat:
        if c["kind"] != "sweep":
            continue
        res = oracle.chain_anchors(c["anchors"], c["gap"], c["score"], want_ops=True)
        assert np.array_equal(res["ops"], np.array(c["ops"], np.int32))
        an = np.array(c["anchors"], np.int64)
        out = np.array(c["out"], np.int32).reshape(-1, 2)
        # every accepted link is the point the reference tree returned for that anchor's query
        starts = [o for o in zip(c["ops"], out) if o[0][0] == 2]
        order = sorted(range(len(an)), key=lambda i: (int(an[i, 0]), i))
        assert len(starts) == len(order)
        for i, (op, (pos, score)) in zip(order, starts):
            if res["prev"][i] != -1:
                assert res["prev"][i] == pos and score != -(1 << 31)


def _random_script(rng, n):
    ncoord, nscore = int(rng.integers(2, 12)), int(rng.integers(1, 5))
    pts = np.stack([rng.integers(0, ncoord, n), np.arange(n)], 1)
    ops, active = [], []
    for i in rng.permutation(n):
        ops.append((0, int(pts[i, 0]), int(i), int(rng.integers(0, nscore)), 0))
        active.append(int(i))
        for _ in range(int(rng.integers(0, 3))):
            a = int(rng.integers(-1, ncoord))
            ops.append((2, a, 0, a + int(rng.integers(0, 6)), n))
        if active and rng.random() < 0.4:
            j = active.pop(int(rng.integers(0, len(active))))
            ops.append((1, int(pts[j, 0]), j, 0, 0))
            a = int(rng.integers(-1, ncoord))
            ops.append((2, a, 0, a + int(rng.integers(0, 6)), n))
    return pts, np.array(ops, np.int32)


def test_oracle_and_host_trees_equal_reference_live(oracle, host, ref_tree):
    rng = np.random.default_rng(7)
    for it in range(400):
        pts, ops = _random_script(rng, int(rng.integers(2, 200)))
        exp = ref_tree.segtree_script(pts, ops)
        for got in (oracle.segtree_script(pts, ops), host.rangemax_script(pts, ops)):
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), it


def _chain_cases(host, rng, n_each):
    cases = [np.zeros((0, 4), np.int32), np.array([[5, 9, 11, 1]], np.int32),
             np.array([[0, 0, 12, 0], [30, 31, 15, 1]], np.int32)]
    for it in range(n_each):  # anchors of mutated copies, every fourth with a tandem repeat (parallel diagonals)
        q = hostgen.rseq(rng, int(rng.integers(50, 4000)), 0.004 if it % 2 else 0.0)
        r = hostgen.rseq(rng, int(rng.integers(0, 300))) + hostgen.mut(rng, q, rng.random() * 0.15) + \
            hostgen.rseq(rng, int(rng.integers(0, 300)))
        if it % 4 == 0:
            rep = hostgen.rseq(rng, int(rng.integers(6, 15))) * 30
            q, r = q[:200] + rep + q[200:], r[:100] + rep + r[100:]
        cases.append(np.array(host.anchors(q, r, 11), np.int32).reshape(-1, 4))
    for it in range(n_each):  # lattices: equal lengths, equal coordinates, equal scores
        m, step = int(rng.integers(1, 300)), int(rng.choice([1, 5, 10]))
        span = int(rng.integers(20, 3000 if it % 2 else 400))
        cases.append(np.stack([rng.integers(0, span // step + 1, m) * step, rng.integers(0, span // step + 1, m) * step,
                               rng.integers(11, 14, m), rng.integers(0, 2, m)], 1).astype(np.int32))
    return cases


def test_oracle_chains_satisfy_the_definition(oracle, host):
    """The oracle's sweep against the O(n^2) definition: dp values, admissible links, chain order."""
    rng = np.random.default_rng(51)
    tied = 0
    for gap, score in ((210, 4), (50, 3)):
        for a in _chain_cases(host, rng, 14):
            chk = bruteforce.ChainCheck(a, gap, score)
            res = oracle.chain_anchors(a, gap, score)
            assert np.array_equal(res["dp"], chk.dp)
            chk.check(res["path"], res["bounds"])
            tied += chk.ties
    assert tied >= 10  # inputs on which the brute force alone could not name the expected output


def test_host_chains_equal_oracle(oracle, host):
    rng = np.random.default_rng(52)
    n = 0
    for gap, score in ((210, 4), (50, 3), (1000, 5)):
        for a in _chain_cases(host, rng, 20):
            res = oracle.chain_anchors(a, gap, score)
            path, bounds = host.chain_raw(a, gap, score)
            assert np.array_equal(path, res["path"]) and np.array_equal(bounds, res["bounds"]), (len(a), gap, score)
            n += 1
    assert n >= 120


@pytest.mark.gpu
def test_device_tree_equals_reference_golden(kat):
    import ctypes as C

    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    f = eng.lib.sdf_debug_chain_tree_script
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    for c in kat:
        pts = np.ascontiguousarray(c["pts"], np.int32)
        ops = np.ascontiguousarray(c["ops"], np.int32).reshape(-1, 5)
        out = np.zeros((max(len(ops), 1), 2), np.int32)
        state = np.zeros(len(c["state"]), np.int32)
        size = f(eng.ctx, pts.ctypes.data, len(pts), ops.ctypes.data, len(ops), out.ctypes.data, state.ctypes.data, len(state))
        assert size == len(c["state"])
        assert _same((out[:len(ops)], state), c), c["kind"]


@pytest.mark.gpu
def test_gpu_chains_equal_oracle(oracle, host):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    rng = np.random.default_rng(53)
    cases = _chain_cases(host, rng, 24)
    for gap, score in ((210, 4), (50, 3)):
        got = eng.chain_batch(cases, gap, score)
        for a, (gp, gb) in zip(cases, got):
            res = oracle.chain_anchors(a, gap, score)
            assert np.array_equal(gp, res["path"]) and np.array_equal(gb, res["bounds"]), (len(a), gap, score)
