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


# ======================================================================================================================
# StageModel: what the reference does BETWEEN its calls into Alignment / Hit / SegmentTree on the way from a bucket
# line to BEDPE lines -- chain_anchors' sweep (src/chain.cc:103-199), fast_align's chain filter and Hit set-up
# (src/chain.cc:203-268), refine_chains (src/refine.cc:23-193) and the stage driver's ordering and remap
# (src/align_main.cc:200-337) -- written from those reference lines as plain control flow.  Every object the reference
# would build there IS the reference's: alignments, merges, guide alignments, to_bed and the priority search tree are
# calls into the reference's own classes compiled unmodified (oracle/_ref/libref_align.so through
# oracle.binding.ReferenceAlign: oracle/ref_hit_table.cc).  Seed anchors come from the brute-force definition above.
# The three reference files themselves cannot be compiled in this image (Boost.ICL); this is the independent check
# their restatement in sedef_amd/csrc/host/pipeline.cc is held to (tests/golden/make_golden_stage_pairs.py).
# ======================================================================================================================
import math


class Ambiguous(Exception):
    """The reference's result depends on something it does not define (an unstable sort over equal keys, n < 2 tree)."""


def c_atoi(s):
    s = s.lstrip(" \t\n\v\f\r")
    i, sign = 0, 1
    if s[:1] in ("+", "-"):
        sign = -1 if s[0] == "-" else 1
        i = 1
    j = i
    while j < len(s) and s[j].isdigit() and s[j].isascii():
        j += 1
    return sign * int(s[i:j]) if j > i else 0


def c_split(s, delim="\t"):
    """split() of src/util.cc:33-41: getline on a stringstream -- a trailing empty field does not exist."""
    parts = s.split(delim)
    if parts and parts[-1] == "":
        parts.pop()
    return parts


_RC = {"A": "T", "a": "t", "C": "G", "c": "g", "G": "C", "g": "c", "T": "A", "t": "a"}


def rc_model(s):
    """rc() (src/util.cc:43-48) with rev_dna (src/common.h:72-87): case kept, everything else becomes 'N'."""
    return "".join(_RC.get(c, "N") for c in reversed(s))


class SeedHit:
    """The fields Hit::from_bed fills (src/hit.cc:29-63)."""

    def __init__(self, line):
        ss = c_split(line)
        assert len(ss) >= 10
        self.qname, self.q_rc = ss[0], ss[8][:1] != "+"
        self.rname, self.r_rc = ss[3], ss[9][:1] != "+"
        self.qs, self.qe, self.rs, self.re = c_atoi(ss[1]), c_atoi(ss[2]), c_atoi(ss[4]), c_atoi(ss[5])
        self.name = ss[6]
        self.comment = ss[14] if len(ss) >= 15 else ""
        self.jaccard = c_atoi(ss[13]) if len(ss) >= 14 else 0


class StageModel:
    # Globals (src/globals.h:71-87, src/globals.cc:20-30), computed the way the reference's initialisers compute them
    MAX_ERROR = 0.30
    MIN_READ_SIZE = int(1000 * (1 - 0.30))
    MAX_CHAIN_GAP = int(0.30 * MIN_READ_SIZE)
    MIN_UPPERCASE_MATCH, MATCH_CHAIN_SCORE = 90, 4
    R_MATCH, R_MISMATCH, R_GAP, R_GAPOPEN = 10.0, 1.0, 0.5, 100.0
    R_MIN_READ, R_SIDE_ALIGN, R_MAX_GAP = 900, 500, 10 * 1000

    def __init__(self, ref):
        self.ref = ref  # oracle.binding.ReferenceAlign
        self.notes = {}

    def note(self, what):
        self.notes[what] = self.notes.get(what, 0) + 1

    # ---- chain_anchors (src/chain.cc:103-199) on the reference's own tree ----
    def chain_anchors(self, anchors):
        n = len(anchors)
        if n == 0:
            return [], [(0, 0)]
        if n == 1:  # SegmentTree's constructor takes clz(n - 1) (src/segment.tpp:18): one anchor is one chain whatever it builds
            return [0], [(0, 0), (1, int(bool(anchors[0][3])))]
        G, MS = self.MAX_CHAIN_GAP, self.MATCH_CHAIN_SCORE
        xs = []
        for i, (q, r, l, hu) in enumerate(anchors):
            xs.append((q, i))
            xs.append((q + l, i))
        max_q = max(a[0] + a[2] for a in anchors)
        max_r = max(a[1] + a[2] for a in anchors)
        max_q, max_r = max(max_q, 0), max(max_r, 0)
        xs.sort()  # Coor::operator< compares x only and x = (coordinate, index) is unique per event kind; a start and an
        #            end event of ONE anchor cannot collide (l > 0)
        self.ref.tree_new([(a[1] + a[2] - 1, i) for i, a in enumerate(anchors)])
        prev = [-1] * n
        dp = [0] * n
        bound = 0
        for k, (x, i) in enumerate(xs):
            q, r, l, hu = anchors[i]
            if x == q:
                while bound < k:
                    t = xs[bound][1]
                    tq, tr, tl, _ = anchors[t]
                    if xs[bound][0] == tq + tl:
                        if q - (tq + tl) <= G:
                            break
                        self.ref.tree_deactivate((tr + tl - 1, t))
                    bound += 1
                w = MS * hu + (MS // 2) * (l - hu)
                j = self.ref.tree_rmq((r - G, 0), (r - 1, n))
                if j >= 0:
                    pq, pr, pl, _ = anchors[j]
                    gap = q - (pq + pl) + r - (pr + pl)
                    if w + dp[j] - gap > 0:
                        dp[i] = w + dp[j] - gap
                        prev[i] = j
                    else:
                        dp[i] = w
                else:
                    dp[i] = w
            else:
                gap = max_q + 1 - (q + l) + max_r + 1 - (r + l)
                self.ref.tree_activate((r + l - 1, i), dp[i] - gap)
        order = sorted(range(n), key=lambda i: (-dp[i], -i))
        path, bounds, used = [], [(0, 0)], [False] * n
        for m in order:
            if used[m]:
                continue
            hu = 0
            while m != -1 and not used[m]:
                path.append(m)
                hu += anchors[m][3]
                used[m] = True
                m = prev[m]
            bounds.append((len(path), int(bool(hu))))
        return path, bounds

    # ---- fast_align (src/chain.cc:203-268) ----
    def fast_align(self, query, ref_seq, orig, kmer):
        """orig: SeedHit-like (qname, rname, q_rc, r_rc, qs, rs).  Returns the handles of the final hits, in order."""
        R = self.ref
        R.tab_open(query, ref_seq)
        same_chr = orig.qname == orig.rname and orig.q_rc == orig.r_rc
        anchors = anchors_bruteforce(query, ref_seq, kmer, same_chr, orig.qs, orig.rs)
        self.last_anchors = anchors
        chain, bounds = self.chain_anchors(anchors)
        self.last_chain = (chain, bounds)
        hits = []
        threshold = self.MIN_READ_SIZE * (1 - self.MAX_ERROR)
        for bi in range(1, len(bounds)):
            has_u = bool(bounds[bi][1])
            be, bs = bounds[bi][0], bounds[bi - 1][0]
            up = int(bounds[bi][1])
            first, last = anchors[chain[be - 1]], anchors[chain[bs]]
            qlo, qhi = first[0], last[0] + last[2]
            rlo, rhi = first[1], last[1] + last[2]
            span = max(rhi - rlo, qhi - qlo)
            if (not has_u or span < self.MIN_UPPERCASE_MATCH) and span < threshold:
                self.note("chain_filtered")
                if span >= 480:
                    self.note("threshold_drop_%d" % span)
                continue
            if (not has_u or span < self.MIN_UPPERCASE_MATCH) and span < 500:
                self.note("threshold_keep_%d" % span)
            guide = [chain[k] for k in range(be - 1, bs - 1, -1)]
            hits.append(R.tab_chain_hit(anchors, guide, qlo, qhi, rlo, rhi, up))
        self.last_first_round = list(hits)
        return self.refine_chains(hits, orig, same_chr)

    # ---- refine_chains (src/refine.cc:23-193) ----
    def refine_chains(self, ids, orig, same_chr):
        R = self.ref
        G = {h: R.tab_get(h) for h in ids}
        key = lambda h: (G[h]["qs"], G[h]["qe"], G[h]["rs"], G[h]["re"])
        if len(set(key(h) for h in ids)) != len(ids):
            raise Ambiguous("std::sort over hits with equal (qs, qe, rs, re) (src/refine.cc:27)")
        A = sorted(ids, key=key)
        n = len(A)
        score = [int(self.R_MATCH * G[h]["matches"] - self.R_MISMATCH * G[h]["mismatches"] - self.R_GAP * G[h]["gap_bases"])
                 for h in A]
        dp, prev, maxes = [0] * n, [-1] * n, []
        oq, orr = orig.qs, orig.rs
        for ai in range(n):
            c = G[A[ai]]
            if same_chr:
                qlo, qhi, rlo, rhi = c["qs"], c["qe"], c["rs"], c["re"]
                qo = max(0, min(oq + qhi, orr + rhi) - max(oq + qlo, orr + rlo))
                if (rhi - rlo) - qo < self.R_SIDE_ALIGN and (qhi - qlo) - qo < self.R_SIDE_ALIGN:
                    self.note("refine_self_overlap_skip")
                    continue
            dp[ai] = score[ai]
            for aj in range(ai - 1, -1, -1):
                p = G[A[aj]]
                cqs = max(c["qs"], p["qe"])
                crs = max(c["rs"], p["re"])
                if p["qe"] >= c["qe"] or p["re"] >= c["re"]:
                    continue
                if p["rs"] >= c["rs"]:
                    continue
                ma = max(cqs - p["qe"], crs - p["re"])
                mi = min(cqs - p["qe"], crs - p["re"])
                if ma >= self.R_MAX_GAP:
                    continue
                if same_chr:
                    qlo, qhi, rlo, rhi = p["qe"], cqs, p["re"], crs
                    qo = max(0, min(oq + qhi, orr + rhi) - max(oq + qlo, orr + rlo))
                    if qo >= 1:
                        self.note("refine_self_overlap_gap")
                        continue
                mis = int(self.R_MISMATCH * mi)
                gap = int(self.R_GAPOPEN + self.R_GAP * (ma - mi))
                sco = dp[aj] + score[ai] - mis - gap
                if sco >= dp[ai]:
                    if sco == dp[ai]:
                        self.note("refine_dp_tie")
                    dp[ai] = sco
                    prev[ai] = aj
            maxes.append((dp[ai], ai))
        maxes = sorted(set(maxes), reverse=True)  # set<pair<int,int>, greater<>>
        used = [False] * n
        hits = []
        for (d, maxi) in maxes:
            if d == 0:
                break
            if used[maxi]:
                continue
            path = []
            while maxi != -1 and not used[maxi]:
                path.insert(0, maxi)
                used[maxi] = True
                maxi = prev[maxi]
            g = lambda k: R.tab_get(A[k])
            first, last = g(path[0]), g(path[-1])
            qlo, qhi, rlo, rhi = first["qs"], last["qe"], first["rs"], last["re"]
            est = first["span"]
            for i in range(1, len(path)):
                cur, prv = g(path[i]), g(path[i - 1])
                est += cur["span"]
                est += max(cur["qs"] - prv["qe"], cur["rs"] - prv["re"])
            if est < self.R_MIN_READ - self.R_SIDE_ALIGN:
                self.note("refine_est_size_drop")
                continue
            overlap = False
            for h in hits:
                f = R.tab_get(h)
                qo = max(0, min(qhi, f["qe"]) - max(qlo, f["qs"]))
                ro = max(0, min(rhi, f["re"]) - max(rlo, f["rs"]))
                if qhi - qlo - qo < self.R_SIDE_ALIGN and rhi - rlo - ro < self.R_SIDE_ALIGN:
                    overlap = True
                    break
            if overlap:
                self.note("refine_overlap_drop")
                continue
            guide = []
            pv = A[path[0]]
            for pi in range(1, len(path)):
                cur = A[path[pi]]
                c, p = R.tab_get(cur), R.tab_get(pv)
                if c["qs"] < p["qe"] or c["rs"] < p["re"]:
                    R.tab_merge(pv, cur)
                    self.note("refine_merge")
                else:
                    guide.append(pv)
                    pv = cur
            guide.append(pv)
            if len(guide) > 1:
                self.note("refine_guide_multi")
            hit = R.tab_guide_hit(guide, qlo, qhi, rlo, rhi, self.R_SIDE_ALIGN)
            if R.tab_get(hit)["span"] >= self.R_MIN_READ:
                hits.append(hit)
            else:
                self.note("refine_final_size_drop")
        return hits

    # ---- generate_alignments (src/align_main.cc:200-337) ----
    @staticmethod
    def schedule(seeds):
        """bucket_alignments(bed, nbins = 1, "", extend = false): complexity classes ascending, file order within."""
        if not seeds:
            return []
        cx = [int(math.sqrt(float(h.qe - h.qs) * float(h.re - h.rs))) for h in seeds]
        bins = [[] for _ in range(max(max(cx), 0) // 1000 + 1)]
        for h, c in zip(seeds, cx):
            bins[c // 1000].append(h)
        return [h for b in bins for h in b]

    def generate(self, fasta_path, fai_text, bed_text, kmer):
        """fasta_path: a FASTA WITHOUT a .fai next to it (see ReferenceAlign.fasta_get); fai_text: the index."""
        index = {}
        for line in fai_text.split("\n"):
            if not line:
                continue
            f = c_split(line)
            assert len(f) == 5
            index[c_split(f[0], " ")[0]] = (f[0], c_atoi(f[1]), int(f[2]), c_atoi(f[3]), c_atoi(f[4]))
        seeds = [SeedHit(s) for s in bed_text.split("\n") if s != ""]
        out = []
        for h in self.schedule(seeds):
            e = index[h.qname]
            fa, h.qe = self.ref.fasta_get(fasta_path, e[0], e[1], e[2], e[3], e[4], h.qs, h.qe)
            e = index[h.rname]
            fb, h.re = self.ref.fasta_get(fasta_path, e[0], e[1], e[2], e[3], e[4], h.rs, h.re)
            if h.r_rc:
                fb = rc_model(fb)
            for hh in self.fast_align(fa, fb, h, kmer):
                g = self.ref.tab_get(hh)
                qs, qe = g["qs"] + h.qs, g["qe"] + h.qs
                if h.r_rc:
                    rs, re_ = g["re"], g["rs"]  # swap
                    rs = h.re - rs
                    re_ = h.re - re_
                else:
                    rs, re_ = g["rs"] + h.rs, g["re"] + h.rs
                self.ref.tab_remap(hh, qs, qe, rs, re_, h.qname, h.rname, h.r_rc)
                out.append(self.ref.tab_to_bed(hh) + "\t" +
                           self.ref.seed_to_bed(h.qname, h.q_rc, h.qs, h.qe, h.rname, h.r_rc, h.rs, h.re, h.name, h.comment,
                                                h.jaccard))
        return out
