"""ctypes bindings for the TEST-ONLY CPU oracle (and, when built, the reference kernel).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (sedef_amd) never imports this module.

  Oracle()      -> oracle/liboracle_extz2.so   (our scalar restatement, extz2_oracle.c)
  Reference()   -> oracle/_ref/libksw2_ref.so  (reference extern/ksw2_extz2_sse.cc compiled as is;
                                                present only where `make -C oracle ref` was run)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NEG_INF = -0x40000000

# SEDEF's scoring (reference: src/align.cc:41-44, src/globals.cc:25-28)
def sedef_mat(match=5, mismatch=-4):
    a, b = match, mismatch if mismatch < 0 else -mismatch
    return np.array([a, b, b, b, 0, b, a, b, b, 0, b, b, a, b, 0, b, b, b, a, 0, 0, 0, 0, 0, 0],
                    dtype=np.int8)


class _OracleResult(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int32), ("max_q", C.c_int32),
                ("max_t", C.c_int32), ("mqe", C.c_int32), ("mqe_t", C.c_int32),
                ("mte", C.c_int32), ("mte_q", C.c_int32), ("score", C.c_int32),
                ("n_cigar", C.c_int64), ("cigar", C.POINTER(C.c_uint32))]


class _KswExtz(C.Structure):  # ksw_extz_t, reference extern/ksw2.h:22-30
    _fields_ = [("max_zd", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int),
                ("score", C.c_int), ("cigar", C.POINTER(C.c_uint32)),
                ("m_cigar", C.c_int64), ("n_cigar", C.c_int64)]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def build_oracle(force=False):
    so = os.path.join(_HERE, "liboracle_extz2.so")
    srcs = [os.path.join(_HERE, f) for f in ("extz2_oracle.c", "stats_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_extz2.so"],
                              stdout=subprocess.DEVNULL)
    return so


def build_reference():
    """Compile the reference kernel from /root/reference if that checkout exists. Returns path or None."""
    so = os.path.join(_HERE, "_ref", "libksw2_ref.so")
    if os.path.exists(so):
        return so
    if os.path.exists("/root/reference/extern/ksw2_extz2_sse.cc"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        return so
    return None


def _as_dict(max_, zd, r, n_cigar, cigar_ptr):
    cig = np.ctypeslib.as_array(cigar_ptr, shape=(n_cigar,)).copy() if n_cigar else \
        np.zeros(0, np.uint32)
    return dict(max=int(max_), zdropped=int(zd), max_q=r.max_q, max_t=r.max_t, mqe=r.mqe,
                mqe_t=r.mqe_t, mte=r.mte, mte_q=r.mte_q, score=r.score, cigar=cig)


def _segtree_script(fn, pts, ops):
    pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 5)
    out = np.zeros((max(len(ops), 1), 2), np.int32)
    cap = 4 << max(1, int(len(pts) - 1).bit_length())
    state = np.zeros(cap, np.int32)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    size = fn(pts.ctypes.data, len(pts), ops.ctypes.data, len(ops), out.ctypes.data, state.ctypes.data, cap)
    assert 0 < size <= cap, size
    return out[:len(ops)].copy(), state[:size].copy()


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.sdfo_extz2.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8),
                                 C.c_int, C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.POINTER(_OracleResult)]
        L.sdfo_extz2.restype = None
        L.sdfo_band_cells.argtypes = [C.c_int, C.c_int, C.c_int]
        L.sdfo_band_cells.restype = C.c_int64
        L.sdfo_extz2_batch.restype = C.c_int64
        L.sdfo_extz2_batch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.POINTER(C.c_int8), C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.sdfo_cigar_counts.restype = None
        L.sdfo_cigar_counts.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]

    def segtree_script(self, pts, ops):
        """chain_oracle.c: replays activate / deactivate / rmq calls on the restated SegmentTree.  pts: n x 2 keys;
        ops: k x 5 (see oracle/ref_align_driver.cc: ref_segtree_script).  Returns (out [k, 2], p pointers of all nodes)."""
        return _segtree_script(self.lib.sdfo_segtree_script, pts, ops)

    def chain_anchors(self, anchors, max_chain_gap=210, match_chain_score=4, want_ops=False):
        """chain_oracle.c: chain_anchors (src/chain.cc:103-199).  anchors: n x (q, r, l, has_u).  Returns a dict with
        path, bounds [(pos, any_upper)], dp, prev [, ops: the tree calls of the sweep]."""
        a = np.ascontiguousarray(anchors, dtype=np.int32).reshape(-1, 4)
        n = len(a)
        path = np.zeros(max(n, 1), np.int32)
        bounds = np.zeros(2 * (n + 1), np.int32)
        nb, nops = C.c_int(0), C.c_int(0)
        dp, prev = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        ops = np.zeros((4 * n + 1, 5), np.int32) if want_ops else None
        f = self.lib.sdfo_chain_anchors
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                      C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        rc = f(a.ctypes.data, n, max_chain_gap, match_chain_score, path.ctypes.data, bounds.ctypes.data, C.byref(nb),
               dp.ctypes.data, prev.ctypes.data, ops.ctypes.data if want_ops else None, C.byref(nops))
        assert rc == 0
        d = dict(path=path[:n].copy(), bounds=bounds[:2 * nb.value].reshape(-1, 2).copy(), dp=dp[:n].copy(),
                 prev=prev[:n].copy())
        if want_ops:
            d["ops"] = ops[:nops.value].copy()
        return d

    def stats_columns(self, a, b, cigar, want_columns=False):
        """stats_oracle.c: a, b FASTA characters (str / bytes), cigar uint32 runs len << 4 | op (0 M, 1 D, 2 I).
        Returns the 16 counters (dict order of STATS_FIELDS) [, align_a, align_b]."""
        a = a.encode() if isinstance(a, str) else bytes(a)
        b = b.encode() if isinstance(b, str) else bytes(b)
        cg = np.ascontiguousarray(cigar, dtype=np.uint32)
        out = np.zeros(16, np.int32)
        ca = C.create_string_buffer(len(a) + len(b) + 1) if want_columns else None
        cb = C.create_string_buffer(len(a) + len(b) + 1) if want_columns else None
        self.lib.sdfo_stats_columns.restype = C.c_int
        self.lib.sdfo_stats_columns.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_char_p, C.c_char_p]
        self.lib.sdfo_stats_columns(a, len(a), b, len(b), cg.ctypes.data, len(cg), out.ctypes.data, ca, cb)
        if want_columns:
            return out, ca.value.decode("latin1"), cb.value.decode("latin1")
        return out

    def extz2(self, query, target, mat=None, m=5, gapo=40, gape=1, w=-1, zdrop=-1, flag=0):
        mat = sedef_mat() if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
        q, qp = _u8(query)
        t, tp = _u8(target)
        r = _OracleResult()
        self.lib.sdfo_extz2(len(q), qp, len(t), tp, m, mat.ctypes.data_as(C.POINTER(C.c_int8)),
                            gapo, gape, w, zdrop, flag, C.byref(r))
        d = _as_dict(r.max, r.zdropped, r, r.n_cigar, r.cigar)
        if r.n_cigar or r.cigar:
            _libc.free(C.cast(r.cigar, C.c_void_p))
        return d

    def band_cells(self, qlen, tlen, w):
        return int(self.lib.sdfo_band_cells(qlen, tlen, w))

    def counts(self, cigar, query, target):
        cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        q, _ = _u8(query)
        t, _ = _u8(target)
        o = [C.c_int32() for _ in range(4)]
        self.lib.sdfo_cigar_counts(cigar.ctypes.data, len(cigar), q.ctypes.data, t.ctypes.data,
                                   *[C.byref(x) for x in o])
        return dict(matches=o[0].value, mismatches=o[1].value, gaps=o[2].value,
                    gap_bases=o[3].value)

    def batch(self, pool, q_off, qlen, t_off, tlen, mat=None, m=5, gapo=40, gape=1, w=-1,
              zdrop=-1, flag=0):
        mat = sedef_mat() if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        t_off = np.ascontiguousarray(t_off, dtype=np.int64)
        qlen = np.ascontiguousarray(qlen, dtype=np.int32)
        tlen = np.ascontiguousarray(tlen, dtype=np.int32)
        n = len(qlen)
        score = np.zeros(n, np.int32)
        h = np.zeros(n, np.uint64)
        self.lib.sdfo_extz2_batch(n, pool.ctypes.data, q_off.ctypes.data, qlen.ctypes.data,
                                  t_off.ctypes.data, tlen.ctypes.data, m,
                                  mat.ctypes.data_as(C.POINTER(C.c_int8)), gapo, gape, w, zdrop, flag,
                                  score.ctypes.data, h.ctypes.data)
        return score, h


class Reference:
    """The reference's own ksw_extz2_sse (extern/ksw2.h:50), compiled unmodified."""

    def __init__(self):
        so = build_reference()
        if so is None:
            raise FileNotFoundError("oracle/_ref/libksw2_ref.so not built and /root/reference absent")
        self.lib = C.CDLL(so)
        f = self.lib.ksw_extz2_sse
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8),
                      C.c_int8, C.POINTER(C.c_int8), C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int,
                      C.POINTER(_KswExtz)]

        b = self.lib.ref_extz2_batch
        b.restype = C.c_int64
        b.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                      C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                      C.c_void_p]

    @staticmethod
    def available():
        return build_reference() is not None

    def batch(self, pool, q_off, qlen, t_off, tlen, mat=None, m=5, gapo=40, gape=1, w=-1,
              zdrop=-1, flag=0):
        """Loops ksw_extz2_sse over the tasks in C (GIL released); returns (scores, cigar hashes)."""
        mat = sedef_mat() if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        t_off = np.ascontiguousarray(t_off, dtype=np.int64)
        qlen = np.ascontiguousarray(qlen, dtype=np.int32)
        tlen = np.ascontiguousarray(tlen, dtype=np.int32)
        n = len(qlen)
        score = np.zeros(n, np.int32)
        h = np.zeros(n, np.uint64)
        self.lib.ref_extz2_batch(n, pool.ctypes.data, q_off.ctypes.data, qlen.ctypes.data,
                                 t_off.ctypes.data, tlen.ctypes.data, m,
                                 mat.ctypes.data_as(C.POINTER(C.c_int8)), gapo, gape, w, zdrop, flag,
                                 score.ctypes.data, h.ctypes.data)
        return score, h

    def extz2(self, query, target, mat=None, m=5, gapo=40, gape=1, w=-1, zdrop=-1, flag=0):
        mat = sedef_mat() if mat is None else np.ascontiguousarray(mat, dtype=np.int8)
        q, qp = _u8(query)
        t, tp = _u8(target)
        r = _KswExtz()
        self.lib.ksw_extz2_sse(None, len(q), qp, len(t), tp, m,
                               mat.ctypes.data_as(C.POINTER(C.c_int8)), gapo, gape, w, zdrop, flag,
                               C.byref(r))
        d = _as_dict(r.max_zd & 0x7fffffff, r.max_zd >> 31, r, r.n_cigar, r.cigar)
        if r.cigar:
            _libc.free(C.cast(r.cigar, C.c_void_p))
        return d


STATS_FIELDS = ("indel_a", "indel_b", "aln_b", "match_b", "mismatch_b", "transitions_b", "transversions_b",
                "uppercase_a", "uppercase_b", "uppercase_matches", "matches", "mismatches", "gaps", "gap_bases", "span",
                "flags")


def cigar_to_str(cigar):
    return "".join("%d%s" % (c >> 4, "MID"[c & 0xf]) for c in np.asarray(cigar).tolist())


# ---- deterministic synthetic inputs (BASELINE config 2 mutation model: 6 % substitution draws,
#      2 % deletions, 2 % insertions; mirrors the reference's simulation idea, python/simulations.py:53-75)
def random_codes(rng, n, n_frac=0.0):
    s = rng.integers(0, 4, size=n, dtype=np.uint8)
    if n_frac > 0:
        s[rng.random(n) < n_frac] = 4
    return s


def mutate(rng, s, sub=0.06, dele=0.02, ins=0.02):
    out = []
    r = rng.random(len(s))
    for k, c in enumerate(s.tolist()):
        x = r[k]
        if x < sub:
            out.append(int(rng.integers(0, 4)))
        elif x < sub + dele:
            continue
        elif x < sub + dele + ins:
            out.append(c)
            out.append(int(rng.integers(0, 4)))
        else:
            out.append(c)
    if not out:
        out = [0]
    return np.array(out, dtype=np.uint8)


# ---- reference Alignment / Hit classes (oracle/_ref/libref_align.so; built by `make -C oracle refalign`) ----
def build_reference_align():
    so = os.path.join(_HERE, "_ref", "libref_align.so")
    have_ref = os.path.exists("/root/reference/src/align.cc")
    drv = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ref_align_driver.cc", "ref_hit_table.cc"))
    if os.path.exists(so) and not (have_ref and os.path.getmtime(so) < drv):
        return so
    if have_ref:
        subprocess.check_call(["make", "-C", _HERE, "refalign"], stdout=subprocess.DEVNULL)
        return so
    return None


class ReferenceAlign:
    """The reference's own Alignment / Hit code (src/align.cc, src/hit.cc) behind oracle/ref_align_driver.cc.
    Loaded with RTLD_LAZY: `split` and `rc` (src/util.cc, needs Boost) are unresolved and never called."""

    def __init__(self):
        so = build_reference_align()
        if so is None:
            raise FileNotFoundError("oracle/_ref/libref_align.so not built and /root/reference absent")
        dl = C.CDLL(None)
        dl.dlopen.restype = C.c_void_p
        dl.dlopen.argtypes = [C.c_char_p, C.c_int]
        h = dl.dlopen(so.encode(), 1)
        if not h:
            raise OSError("dlopen failed: " + so)
        self.lib = C.CDLL(so, handle=h)
        self.buf = C.create_string_buffer(4 << 20)

    def alignment_pair(self, fa, fb):
        cnt = (C.c_int * 5)()
        rc = self.lib.ref_alignment_pair(fa.encode(), fb.encode(), self.buf, len(self.buf), cnt)
        assert rc == 0
        return self.buf.value.decode(), list(cnt)

    def alignment_from_cigar(self, fa, fb, cigar):
        """Reference Alignment(fa, fb, cigar string): (align_a, align_b, [matches, mismatches, gaps, gap_bases, span])."""
        cnt = (C.c_int * 5)()
        rc = self.lib.ref_alignment_from_cigar(fa.encode(), fb.encode(), cigar.encode(), self.buf, len(self.buf), cnt)
        assert rc == 0
        aa, _, ab = self.buf.value.decode().split("\n")[:3]
        return aa, ab, list(cnt)

    def guide_from_chains(self, q, r, spec, side):
        rc = self.lib.ref_guide_from_chains(q.encode(), r.encode(), spec.encode(), side, self.buf, len(self.buf))
        assert rc == 0
        return self.buf.value.decode()

    def merge(self, spec, merge_dist):
        rc = self.lib.ref_merge(spec.encode(), merge_dist, self.buf, len(self.buf))
        assert rc == 0
        return self.buf.value.decode()

    def fasta_get(self, fasta_path, name, length, offset, line_blen, line_len, start, end=None):
        """FastaReference::get_sequence (src/fasta.cc:105-142); `fasta_path` must have no .fai next to it (the one
        index entry is given here).  Returns (sequence, end after clamping)."""
        e = C.c_int(0 if end is None else end)
        f = self.lib.ref_fasta_get
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        rc = f(fasta_path.encode(), name.encode(), length, offset, line_blen, line_len, start, int(end is not None),
               C.byref(e), self.buf, len(self.buf))
        assert rc == 0, self.buf.value
        return self.buf.value.decode(), (None if end is None else e.value)

    def hit_extend(self, qs, qe, rs, re_, factor=5.0, max_extend=15000):
        io = (C.c_int * 4)(qs, qe, rs, re_)
        self.lib.ref_hit_extend.argtypes = [C.c_void_p, C.c_double, C.c_int]
        self.lib.ref_hit_extend(io, factor, max_extend)
        return list(io)

    def sequence(self, name, seq):
        rc = self.lib.ref_sequence(name.encode(), seq.encode(), self.buf, len(self.buf))
        assert rc == 0
        n, s, r = self.buf.value.decode().split("|")
        return n, s, r == "1"

    def fmt_double(self, x):
        """fmt::format("{}", double) of the reference's vendored fmt (extern/format.cc)."""
        self.lib.ref_fmt_double.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
        assert self.lib.ref_fmt_double(float(x), self.buf, len(self.buf)) == 0
        return self.buf.value.decode()

    def segtree_script(self, pts, ops):
        """The reference's own SegmentTree<T> (src/segment.h, src/segment.tpp) driven by a script of activate /
        deactivate / rmq calls (oracle/ref_align_driver.cc: ref_segtree_script).  At least two points."""
        return _segtree_script(self.lib.ref_segtree_script, pts, ops)

    def set_scoring(self, match=5, mismatch=-4, gap_open=-40, gap_extend=-1):
        """Globals::Align::* as the CLI overrides set them (src/align_main.cc:343-352); process-wide in the reference."""
        self.lib.ref_set_scoring(match, mismatch, gap_open, gap_extend)

    # ---- the reference's Hit / Alignment objects behind handles (oracle/ref_hit_table.cc) ----
    def tab_open(self, query, ref):
        """fast_align's shared sequences (src/chain.cc:207-208); drops the previous pair's handles."""
        assert self.lib.ref_tab_open(query.encode(), ref.encode()) == 0

    def tab_chain_hit(self, anchors, guide, qlo, qhi, rlo, rhi, up):
        """Hit{...}; aln = Alignment(query, ref, anchors, guide); update_from_alignment (src/chain.cc:243-258)."""
        a = np.ascontiguousarray(anchors, dtype=np.int32).reshape(-1, 4)
        g = np.ascontiguousarray(guide, dtype=np.int32)
        f = self.lib.ref_tab_chain_hit
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 5
        h = f(a.ctypes.data, len(a), g.ctypes.data, len(g), qlo, qhi, rlo, rhi, up)
        assert h >= 0
        return h

    def tab_get(self, h):
        """dict of query_start, query_end, ref_start, ref_end, jaccard, matches, mismatches, gap_bases, gaps, span."""
        out = (C.c_int * 10)()
        assert self.lib.ref_tab_get(h, out) == 0
        return dict(zip(("qs", "qe", "rs", "re", "jaccard", "matches", "mismatches", "gap_bases", "gaps", "span"), list(out)))

    def tab_cigar(self, h):
        assert self.lib.ref_tab_cigar(h, self.buf, C.c_size_t(len(self.buf))) == 0
        return self.buf.value.decode()

    def tab_merge(self, prev, cur):
        """prev->aln.merge(cur.aln, qseq, rseq); update_from_alignment(*prev) (src/refine.cc:172-173)."""
        assert self.lib.ref_tab_merge(prev, cur) == 0

    def tab_guide_hit(self, guide, qlo, qhi, rlo, rhi, side):
        """Hit{...}; aln = Alignment(qseq, rseq, guide, side); update_from_alignment (src/refine.cc:164-183)."""
        g = np.ascontiguousarray(guide, dtype=np.int32)
        f = self.lib.ref_tab_guide_hit
        f.argtypes = [C.c_void_p, C.c_int] + [C.c_int] * 5
        h = f(g.ctypes.data, len(g), qlo, qhi, rlo, rhi, side)
        assert h >= 0
        return h

    def tab_remap(self, h, qs, qe, rs, re_, qname, rname, ref_is_rc):
        """The stage driver's in-place update of a result (src/align_main.cc:314-327) with coordinates the caller computed."""
        assert self.lib.ref_tab_remap(h, qs, qe, rs, re_, qname.encode(), rname.encode(), int(ref_is_rc)) == 0

    def tab_to_bed(self, h):
        assert self.lib.ref_tab_to_bed(h, self.buf, C.c_size_t(len(self.buf))) == 0
        return self.buf.value.decode()

    def seed_to_bed(self, qname, q_rc, qs, qe, rname, r_rc, rs, re_, name, comment, jaccard):
        """to_bed(0) of a Hit filled like Hit::from_bed fills it (src/hit.cc:29-63) from fields the caller split."""
        f = self.lib.ref_seed_to_bed
        f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p,
                      C.c_int, C.c_char_p, C.c_size_t]
        assert f(qname.encode(), int(q_rc), qs, qe, rname.encode(), int(r_rc), rs, re_, name.encode(), comment.encode(),
                 jaccard, self.buf, len(self.buf)) == 0
        return self.buf.value.decode()

    # ---- the reference's SegmentTree, live (oracle/ref_hit_table.cc) ----
    def tree_new(self, pts):
        p = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
        self.lib.ref_tree_new.argtypes = [C.c_void_p, C.c_int]
        assert self.lib.ref_tree_new(p.ctypes.data, len(p)) == 0

    def tree_activate(self, x, score):
        self.lib.ref_tree_activate(int(x[0]), int(x[1]), int(score))

    def tree_deactivate(self, x):
        self.lib.ref_tree_deactivate(int(x[0]), int(x[1]))

    def tree_rmq(self, p, q):
        """-1: nothing; -2: a point that was never activated (ys[j].score == MIN, src/chain.cc:160); else ys[j].pos."""
        return self.lib.ref_tree_rmq(int(p[0]), int(p[1]), int(q[0]), int(q[1]))
