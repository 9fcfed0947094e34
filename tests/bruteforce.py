"""Independent brute-force checkers for seed anchors and anchor chaining (TEST INFRASTRUCTURE).

Written from the *definition* of what the reference computes, not from its data structures, so that both the host
code (sedef_amd/csrc/host) and the HIP kernels can be checked against something that is not a port of either:

  * anchors_bruteforce: O(|q|·|r|) enumeration of maximal exact matches per diagonal.  The reference
    (src/chain.cc:24-101) streams query k-mers through a hash join and keeps, per diagonal, the query position up to
    which matches are already covered (`slide[]`, src/chain.cc:42,70-99).  Net effect per diagonal and per maximal
    case-insensitive N-free match run [s, e): ONE anchor, starting at the first position of the run whose k-mer is
    eligible (fewer than 1000 occurrences in the reference sequence, src/chain.cc:61; not within k of the main diagonal
    of a same-chromosome pair, src/chain.cc:67-69) and reaching to e; `has_u` is "any uppercase base in either
    sequence" (a bool accumulated with +=, src/chain.cc:74,84).  Order: query position, then reference position.

  * ChainCheck: O(n²) dynamic programme for chain_anchors (src/chain.cc:103-199).  Predecessors of anchor i are the
    anchors t whose end event (t.q+t.l, t) sorts before i's start event (i.q, i) (src/chain.cc:135), with
    0 <= i.q-(t.q+t.l) <= MAX_CHAIN_GAP (src/chain.cc:142-152) and 0 <= i.r-(t.r+t.l) <= MAX_CHAIN_GAP-1
    (src/chain.cc:157-158).  The tree returns a predecessor of maximal stored score dp[t]-(distance of t's end to the
    far corner) (src/chain.cc:175-176); WHICH one among equal scores depends on the priority-search tree's shape and
    activation history (src/segment.tpp:62,89,128), so `check` treats every maximal one as admissible: dp values do
    not depend on the choice (dp[i] = w + stored + const(i)), only the `prev` pointers do.  Checked exactly: dp values
    through the chain order, chain starts (score-descending, index-descending, src/chain.cc:179-186), that each link
    is an admissible predecessor, that each chain stops exactly where no positive predecessor exists or the taken one
    is used.  The tie winner itself is pinned elsewhere: oracle/chain_oracle.c restates the tree, tests/test_chain_oracle.py
    holds it equal to the reference's own SegmentTree class on tie-rich scripts, and `check_exact` compares a result with
    the oracle's chains for equality -- ties included -- after checking the oracle's result against this definition.
"""
import numpy as np


def _codes(s):
    a = np.frombuffer(s.upper().encode(), dtype=np.uint8)
    return a


def anchors_bruteforce(q, r, k, same_chr=False, qstart=0, rstart=0, max_occ=1000):
    n, m = len(q), len(r)
    if n < k or m < k:
        return []
    Q, R = _codes(q), _codes(r)
    qu = np.frombuffer(q.encode(), dtype=np.uint8)
    ru = np.frombuffer(r.encode(), dtype=np.uint8)
    q_upper = (qu >= 65) & (qu <= 90)
    r_upper = (ru >= 65) & (ru <= 90)
    # occurrences of every N-free reference k-mer
    occ = {}
    R_str = r.upper()
    for i in range(m - k + 1):
        w = R_str[i:i + k]
        if "N" not in w:
            occ[w] = occ.get(w, 0) + 1
    Q_str = q.upper()
    q_ok = np.zeros(n, bool)  # k-mer starting here is eligible by frequency
    for i in range(n - k + 1):
        w = Q_str[i:i + k]
        c = occ.get(w, 0)
        q_ok[i] = ("N" not in w) and 0 < c < max_occ
    out = []
    for d in range(-(n - 1), m):  # r = q + d
        if same_chr and abs(rstart + d - qstart) <= k:
            continue
        lo, hi = max(0, -d), min(n, m - d)
        if hi - lo < k:
            continue
        a, b = Q[lo:hi], R[lo + d:hi + d]
        eq = (a == b) & (a != 78)
        if not eq.any():
            continue
        e = np.concatenate(([0], eq.astype(np.int8), [0]))
        starts = np.flatnonzero(np.diff(e) == 1)
        ends = np.flatnonzero(np.diff(e) == -1)
        for s, t in zip(starts.tolist(), ends.tolist()):
            if t - s < k:
                continue
            cand = np.flatnonzero(q_ok[lo + s:lo + t - k + 1])
            if len(cand) == 0:
                continue
            qq = lo + s + int(cand[0])
            ln = lo + t - qq
            hu = bool(q_upper[qq:qq + ln].any() or r_upper[qq + d:qq + d + ln].any())
            out.append((qq, qq + d, ln, int(hu)))
    out.sort(key=lambda x: (x[0], x[1]))
    return out


class ChainCheck:
    def __init__(self, anchors, max_gap=210, match_score=4):
        a = np.asarray(anchors, dtype=np.int64).reshape(-1, 4)
        self.a = a
        n = len(a)
        self.n = n
        self.G = max_gap
        q, r, l, hu = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        self.w = match_score * hu + (match_score // 2) * (l - hu)
        qe, re_ = q + l, r + l
        self.dp = np.zeros(n, np.int64)
        self.opt = [set() for _ in range(n)]  # admissible predecessors (empty: the chain ends here)
        self.ties = False
        order = sorted(range(n), key=lambda i: (int(q[i]), i))
        idx = np.arange(n)
        for i in order:
            before = (qe < q[i]) | ((qe == q[i]) & (idx < i))  # end event sorts before the start event
            el = before & (q[i] - qe <= max_gap) & (re_ <= r[i]) & (r[i] - re_ <= max_gap - 1)
            el[i] = False
            self.dp[i] = self.w[i]
            if el.any():
                cand = np.flatnonzero(el)
                # stored score up to a constant: dp[t] + t.qe + t.re  (the far-corner distance, src/chain.cc:175)
                stored = self.dp[cand] + qe[cand] + re_[cand]
                best = stored.max()
                top = cand[stored == best]
                t = int(top[0])
                val = int(self.w[i] + self.dp[t] - ((q[i] - qe[t]) + (r[i] - re_[t])))
                if val > 0:
                    self.dp[i] = val
                    self.opt[i] = set(int(x) for x in top)
                    if len(top) > 1:
                        self.ties = True

    def expected_if_unique(self):
        """(path, boundaries) when no tie ever occurred, else None."""
        if self.ties:
            return None
        used = np.zeros(self.n, bool)
        path, bounds = [], [(0, 0)]
        for i in sorted(range(self.n), key=lambda i: (-int(self.dp[i]), -i)):
            if used[i]:
                continue
            hu = 0
            while i != -1 and not used[i]:
                path.append(i)
                hu += int(self.a[i, 3])
                used[i] = True
                i = next(iter(self.opt[i])) if self.opt[i] else -1
            bounds.append((len(path), int(bool(hu))))
        return path, bounds

    def check_exact(self, path, bounds, oracle_result):
        """`oracle_result` (oracle.binding.Oracle.chain_anchors: the sweep on the pinned tree) is itself admissible by
        this definition, and (path, bounds) equals it exactly -- with the tie winners the reference's tree picks."""
        assert np.array_equal(oracle_result["dp"], self.dp)
        self.check(oracle_result["path"], oracle_result["bounds"])
        assert np.array_equal(np.asarray(path, np.int64), oracle_result["path"]), "path differs from the oracle's"
        assert np.array_equal(np.asarray(bounds, np.int64).reshape(-1, 2), oracle_result["bounds"]), "boundaries differ"
        return True

    def check(self, path, bounds):
        """Asserts that (path, boundaries) is an output chain_anchors can produce for these anchors."""
        path = [int(x) for x in path]
        bounds = [(int(b[0]), int(b[1])) for b in bounds]
        n = self.n
        assert sorted(path) == list(range(n)), "every anchor appears exactly once"
        assert bounds[0] == (0, 0) and bounds[-1][0] == n if n else bounds == [(0, 0)]
        used = np.zeros(n, bool)
        order = sorted(range(n), key=lambda i: (-int(self.dp[i]), -i))
        oi = 0
        for b in range(1, len(bounds)):
            s, e = bounds[b - 1][0], bounds[b][0]
            assert e > s
            while used[order[oi]]:
                oi += 1
            assert path[s] == order[oi], "chain %d must start at the best unused anchor" % b
            hu = 0
            for k in range(s, e):
                i = path[k]
                assert not used[i]
                used[i] = True
                hu += int(self.a[i, 3])
                if k + 1 < e:
                    assert path[k + 1] in self.opt[i], "link %d -> %d is not a best predecessor" % (i, path[k + 1])
                else:  # the chain stops: no positive predecessor, or the one taken is already used
                    assert not self.opt[i] or any(used[t] for t in self.opt[i]), \
                        "chain stopped at %d although its predecessor is free" % i
            assert bounds[b][1] == int(bool(hu)), "any-uppercase flag of chain %d" % b
        return True
