"""ctypes mirror of include/sedef_hip.h.

`ksw_extz2()` keeps the argument order and meaning of the reference's ksw_extz2_sse
(reference: extern/ksw2.h:50) so parity tests read like calls to the reference;
`Extz2Engine` exposes the batched entry points.
"""
import ctypes as C
import os

import numpy as np

from .build import LIB_PATH

NEG_INF = -0x40000000
WANT_CIGAR, WANT_SCORE, WANT_EXT, WANT_ALL = 1, 2, 4, 7

TASK_DTYPE = np.dtype([("q_off", "<i8"), ("t_off", "<i8"), ("qlen", "<i4"), ("tlen", "<i4"),
                       ("w", "<i4"), ("zdrop", "<i4"), ("flag", "<i4"), ("pad_", "<i4")])
RESULT_DTYPE = np.dtype([("score", "<i4"), ("max", "<i4"), ("max_q", "<i4"), ("max_t", "<i4"),
                         ("mqe", "<i4"), ("mqe_t", "<i4"), ("mte", "<i4"), ("mte_q", "<i4"),
                         ("zdropped", "<i4"), ("n_cigar", "<i4"), ("cigar_off", "<i8"),
                         ("matches", "<i4"), ("mismatches", "<i4"), ("gaps", "<i4"),
                         ("gap_bases", "<i4")])
ANCHOR_PAIR_DTYPE = np.dtype([("q_off", "<i8"), ("r_off", "<i8"), ("qlen", "<i4"), ("rlen", "<i4"),
                              ("same_chr", "<i4"), ("delta", "<i4")])
ANCHOR_DTYPE = np.dtype([("q", "<i4"), ("r", "<i4"), ("l", "<i4"), ("has_u", "<i4")])
BRIEF_DTYPE = np.dtype([("cigar_off", "<i8"), ("n_cigar", "<i4"), ("matches", "<i4")])  # sdf_result_brief
RESERVE_BRIEF, RESERVE_ANCHORS = 1, 2
assert BRIEF_DTYPE.itemsize == 16
assert TASK_DTYPE.itemsize == 40 and RESULT_DTYPE.itemsize == 64 and ANCHOR_PAIR_DTYPE.itemsize == 32
# include/sedef_hip.h: sdf_stats_task / sdf_stats_cols
STATS_TASK_DTYPE = np.dtype([("a_off", "<u8"), ("b_off", "<u8"), ("a_len", "<u4"), ("b_len", "<u4"),
                             ("cigar_off", "<u8"), ("n_cigar", "<u4"), ("reserved", "<u4")])
STATS_COLS_DTYPE = np.dtype([(f, "<i4") for f in (
    "indel_a", "indel_b", "aln_b", "match_b", "mismatch_b", "transitions_b", "transversions_b", "uppercase_a",
    "uppercase_b", "uppercase_matches", "matches", "mismatches", "gaps", "gap_bases", "span", "flags")])
assert STATS_TASK_DTYPE.itemsize == 40 and STATS_COLS_DTYPE.itemsize == 64


class SdfError(RuntimeError):
    pass


class _Scoring(C.Structure):
    _fields_ = [("m", C.c_int32), ("mat", C.c_int8 * 25), ("gapo", C.c_int8), ("gape", C.c_int8),
                ("pad_", C.c_int8)]


_lib = None


def library_path():
    return LIB_PATH


def load_library():
    """Loads the HIP library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdfError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                       % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.sdf_device_count.restype = C.c_int
    L.sdf_create.restype = C.c_void_p
    L.sdf_create.argtypes = [C.c_int, C.c_size_t]
    L.sdf_destroy.argtypes = [C.c_void_p]
    L.sdf_last_error.restype = C.c_char_p
    L.sdf_last_error.argtypes = [C.c_void_p]
    L.sdf_packed_words.restype = C.c_size_t
    L.sdf_packed_words.argtypes = [C.c_int32]
    L.sdf_pack_codes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.sdf_band_cells.restype = C.c_int64
    L.sdf_band_cells.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.sdf_extz2_batch.restype = C.c_int
    L.sdf_extz2_batch.argtypes = [C.c_void_p, C.POINTER(_Scoring), C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdf_extz2_batch_brief.restype = C.c_int
    L.sdf_extz2_batch_brief.argtypes = [C.c_void_p, C.POINTER(_Scoring), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdf_pool_host.restype = C.c_void_p
    L.sdf_pool_host.argtypes = [C.c_void_p, C.c_size_t]
    L.sdf_pool_upload.restype = C.c_int
    L.sdf_pool_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.sdf_pool_bytes.restype = C.c_size_t
    L.sdf_pool_bytes.argtypes = [C.c_void_p]
    L.sdf_extz2_batch_pairs.restype = C.c_int
    L.sdf_extz2_batch_pairs.argtypes = [C.c_void_p, C.POINTER(_Scoring), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdf_extz2_batch_pairs_full.restype = C.c_int
    L.sdf_extz2_batch_pairs_full.argtypes = [C.c_void_p, C.POINTER(_Scoring), C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdf_reserve.restype = C.c_int
    L.sdf_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32]
    L.sdf_device_bytes.restype = C.c_size_t
    L.sdf_device_bytes.argtypes = [C.c_void_p]
    L.sdf_extz2_batch_device.restype = C.c_int
    L.sdf_extz2_batch_device.argtypes = [C.c_void_p, C.POINTER(_Scoring), C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.sdf_anchors_batch.restype = C.c_int
    L.sdf_anchors_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.sdf_chain_batch.restype = C.c_int
    L.sdf_chain_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    L.sdf_stats_columns_batch.restype = C.c_int
    L.sdf_stats_columns_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p,
                                          C.c_size_t, C.c_void_p]
    L.sdf_stats_columns_device.restype = C.c_int
    L.sdf_stats_columns_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    L.sdf_last_ms.restype = C.c_float
    L.sdf_last_ms.argtypes = [C.c_void_p, C.c_int]
    L.sdf_last_launches.restype = C.c_int
    L.sdf_last_launches.argtypes = [C.c_void_p]
    L.sdf_last_paired.restype = C.c_longlong
    L.sdf_last_paired.argtypes = [C.c_void_p]
    L.sdf_last_reran.restype = C.c_longlong
    L.sdf_last_reran.argtypes = [C.c_void_p]
    L.sdf_last_lane_tasks.restype = C.c_longlong
    L.sdf_last_lane_tasks.argtypes = [C.c_void_p]
    _lib = L
    return L


def packed_words(n):
    return int(load_library().sdf_packed_words(int(n)))


def pack_codes(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(packed_words(len(codes)), np.uint32)
    if len(codes):
        load_library().sdf_pack_codes(codes.ctypes.data, len(codes), out.ctypes.data)
    return out


def band_cells(qlen, tlen, w):
    return int(load_library().sdf_band_cells(qlen, tlen, w))


def _scoring(mat, gapo, gape, m=5):
    s = _Scoring()
    s.m = m
    mat = np.asarray(mat, dtype=np.int8)
    for i in range(min(25, len(mat))):
        s.mat[i] = int(mat[i])
    s.gapo, s.gape = gapo, gape
    return s


def sedef_mat(match=5, mismatch=-4):
    """The 5x5 matrix align_helper builds (reference: src/align.cc:41-44)."""
    a, b = match, mismatch if mismatch < 0 else -mismatch
    return np.array([a, b, b, b, 0, b, a, b, b, 0, b, b, a, b, 0, b, b, b, a, 0, 0, 0, 0, 0, 0],
                    dtype=np.int8)


class Config:
    """include/sedef_hip.h: sdf_config -- the library's settings as one opaque struct.  Config() holds what sdf_create would
    read from the environment NOW; Config(SDF_NO_PAIR=1, strip_cols=4) sets fields by their environment variable's or
    their own name (a wrong name or a value out of range raises); nothing is written to os.environ."""

    BYTES = 512  # (room for the struct: its first word is its size, checked below)

    def __init__(self, from_env=True, **settings):
        self.lib = load_library()
        self.buf = C.create_string_buffer(self.BYTES)
        err = C.create_string_buffer(512)
        if from_env:
            self.lib.sdf_config_from_env.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
            if self.lib.sdf_config_from_env(self.buf, err, len(err)) != 0:
                raise SdfError("environment: %s" % err.value.decode())
        else:
            self.lib.sdf_config_default.argtypes = [C.c_void_p]
            self.lib.sdf_config_default(self.buf)
        assert 0 < C.cast(self.buf, C.POINTER(C.c_uint32))[0] <= self.BYTES
        self.set(**settings)

    def set(self, **settings):
        err = C.create_string_buffer(512)
        self.lib.sdf_config_set.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        for k, v in settings.items():
            if isinstance(v, bool):
                v = int(v)
            if self.lib.sdf_config_set(self.buf, k.encode(), str(v).encode(), err, len(err)) != 0:
                raise SdfError(err.value.decode())
        return self

    def dump(self):
        self.lib.sdf_config_dump.restype = C.c_size_t
        self.lib.sdf_config_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        out = C.create_string_buffer(self.lib.sdf_config_dump(self.buf, None, 0))
        self.lib.sdf_config_dump(self.buf, out, len(out))
        return out.value.decode()

    def as_dict(self):
        return {k: float(v.split()[0]) for k, v in (ln.split("=", 1) for ln in self.dump().splitlines() if "=" in ln)}


def describe_config():
    lib = load_library()
    lib.sdf_config_describe.restype = C.c_size_t
    lib.sdf_config_describe.argtypes = [C.c_char_p, C.c_size_t]
    out = C.create_string_buffer(lib.sdf_config_describe(None, 0))
    lib.sdf_config_describe(out, len(out))
    return out.value.decode()


class Extz2Engine:
    """One context = one GPU (include/sedef_hip.h: sdf_create / sdf_create_cfg).  config: a Config, or a dict of settings
    on top of the environment's (Extz2Engine(0, config=dict(SDF_NO_PAIR=1)))."""

    def __init__(self, device=0, workspace_bytes=0, config=None):
        self.lib = load_library()
        if config is None:
            self.ctx = self.lib.sdf_create(device, workspace_bytes)
        else:
            cfg = config if isinstance(config, Config) else Config(**config)
            self.lib.sdf_create_cfg.restype = C.c_void_p
            self.lib.sdf_create_cfg.argtypes = [C.c_int, C.c_size_t, C.c_void_p]
            self.ctx = self.lib.sdf_create_cfg(device, workspace_bytes, cfg.buf)
        if not self.ctx:
            raise SdfError("sdf_create failed: %s" % self.lib.sdf_last_error(None).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sdf_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SdfError("rc=%d: %s" % (rc, self.lib.sdf_last_error(self.ctx).decode()))

    def align_pairs(self, pairs, w=-1, zdrop=-1, flag=0, mat=None, gapo=40, gape=1,
                    want=WANT_ALL):
        """pairs: list of (query_codes, target_codes).  w/zdrop/flag: scalar or per-pair list.
        Returns (results structured array, cigar pool)."""
        n = len(pairs)
        tasks = np.zeros(n, TASK_DTYPE)
        chunks, off = [], 0
        for k, (q, t) in enumerate(pairs):
            q = np.ascontiguousarray(q, dtype=np.uint8)
            t = np.ascontiguousarray(t, dtype=np.uint8)
            tasks["q_off"][k], tasks["qlen"][k] = off, len(q)
            off += len(q)
            tasks["t_off"][k], tasks["tlen"][k] = off, len(t)
            off += len(t)
            chunks += [q, t]
        pool = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
        tasks["w"], tasks["zdrop"], tasks["flag"] = w, zdrop, flag
        return self.align_batch(tasks, pool, mat=mat, gapo=gapo, gape=gape, want=want)

    def align_batch(self, tasks, pool, mat=None, gapo=40, gape=1, want=WANT_ALL, cigar_cap=None):
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        n = len(tasks)
        sc = _scoring(sedef_mat() if mat is None else mat, gapo, gape)
        out = np.zeros(n, RESULT_DTYPE)
        if cigar_cap is None:
            cigar_cap = int((tasks["qlen"].astype(np.int64) + tasks["tlen"] + 2).sum()) + 1
        cig = np.zeros(cigar_cap, np.uint32)
        used = C.c_size_t(0)
        rc = self.lib.sdf_extz2_batch(self.ctx, C.byref(sc), tasks.ctypes.data, n, pool.ctypes.data,
                                      pool.nbytes, want, out.ctypes.data, cig.ctypes.data,
                                      cigar_cap, C.byref(used))
        self._check(rc)
        return out, cig[:used.value]

    def align_batch_brief(self, tasks, pool, mat=None, gapo=40, gape=1, cigar_cap=None):
        """sdf_extz2_batch_brief: 16-byte records (cigar_off, n_cigar, matches) instead of sdf_result."""
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        n = len(tasks)
        sc = _scoring(sedef_mat() if mat is None else mat, gapo, gape)
        out = np.zeros(n, BRIEF_DTYPE)
        if cigar_cap is None:
            cigar_cap = int((tasks["qlen"].astype(np.int64) + tasks["tlen"] + 2).sum()) + 1
        cig = np.zeros(cigar_cap, np.uint32)
        used = C.c_size_t(0)
        self._check(self.lib.sdf_extz2_batch_brief(self.ctx, C.byref(sc), tasks.ctypes.data, n, pool.ctypes.data, pool.nbytes,
                                                   out.ctypes.data, cig.ctypes.data, cigar_cap, C.byref(used)))
        return out, cig[:used.value]

    def pool_upload(self, chars, pinned=True):
        """sdf_pool_host + sdf_pool_upload: raw FASTA characters (bytes) to HBM, where they stay; sdf_extz2_batch_pairs tasks
        name byte ranges of them.  pinned=False: straight from the caller's (pageable) buffer."""
        buf = np.frombuffer(chars, np.uint8)
        if pinned:
            p = self.lib.sdf_pool_host(self.ctx, len(buf))
            if not p:
                raise SdfError(self.lib.sdf_last_error(self.ctx).decode())
            C.memmove(p, buf.ctypes.data, len(buf))
            self._check(self.lib.sdf_pool_upload(self.ctx, p, len(buf)))
        else:
            self._keep = buf  # (unchanged until the next call that returns data)
            self._check(self.lib.sdf_pool_upload(self.ctx, buf.ctypes.data, len(buf)))
        return int(self.lib.sdf_pool_bytes(self.ctx))

    def align_batch_pairs(self, tasks, mat=None, gapo=40, gape=1, want=None, cigar_cap=None):
        """sdf_extz2_batch_pairs (want=None: 16-byte records) / sdf_extz2_batch_pairs_full: q_off / t_off of the tasks are byte
        offsets into the resident character pool; align_dna and the packing happen on the device."""
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        n = len(tasks)
        sc = _scoring(sedef_mat() if mat is None else mat, gapo, gape)
        if cigar_cap is None:
            cigar_cap = int((tasks["qlen"].astype(np.int64) + tasks["tlen"] + 2).sum()) + 1
        cig = np.zeros(cigar_cap, np.uint32)
        used = C.c_size_t(0)
        if want is None:
            out = np.zeros(n, BRIEF_DTYPE)
            self._check(self.lib.sdf_extz2_batch_pairs(self.ctx, C.byref(sc), tasks.ctypes.data, n, out.ctypes.data,
                                                       cig.ctypes.data, cigar_cap, C.byref(used)))
        else:
            out = np.zeros(n, RESULT_DTYPE)
            self._check(self.lib.sdf_extz2_batch_pairs_full(self.ctx, C.byref(sc), tasks.ctypes.data, n, want, out.ctypes.data,
                                                            cig.ctypes.data, cigar_cap, C.byref(used)))
        return out, cig[:used.value]

    def reserve(self, max_tasks, max_bases, workspace_bytes=0, flags=0):
        """sdf_reserve: buffers, pinned staging and pipeline streams sized once (flags: RESERVE_BRIEF | RESERVE_ANCHORS)."""
        self._check(self.lib.sdf_reserve(self.ctx, int(max_tasks), int(max_bases), int(workspace_bytes), int(flags)))

    def device_bytes(self):
        return int(self.lib.sdf_device_bytes(self.ctx))

    def align_batch_device(self, tasks, d_pool, d_out, d_cig, cigar_cap, mat=None, gapo=40, gape=1,
                           want=WANT_ALL, stream=None):
        """Device-resident form: d_pool/d_out/d_cig are raw HBM addresses (ints)."""
        tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
        sc = _scoring(sedef_mat() if mat is None else mat, gapo, gape)
        used = C.c_size_t(0)
        rc = self.lib.sdf_extz2_batch_device(self.ctx, C.byref(sc), tasks.ctypes.data, len(tasks),
                                             d_pool, want, d_out, d_cig, cigar_cap, C.byref(used),
                                             stream)
        self._check(rc)
        return used.value

    def anchors_batch(self, pairs, kmer=11):
        """GPU generate_anchors.  pairs: list of (query str, ref str, same_chr, delta).  Returns one list of
        (q, r, l, has_u) per pair (include/sedef_hip.h: sdf_anchors_batch)."""
        n = len(pairs)
        desc = np.zeros(n, ANCHOR_PAIR_DTYPE)
        chunks, off = [], 0
        for k, (q, r, same, delta) in enumerate(pairs):
            qb, rb = q.encode(), r.encode()
            desc[k] = (off, off + len(qb), len(qb), len(rb), int(same), int(delta))
            off += len(qb) + len(rb)
            chunks += [qb, rb]
        pool = b"".join(chunks)
        cap = max(1024, off)
        while True:
            out = np.zeros(cap, ANCHOR_DTYPE)
            offs = np.zeros(n + 1, np.int64)
            used = C.c_size_t(0)
            rc = self.lib.sdf_anchors_batch(self.ctx, desc.ctypes.data, n, pool, len(pool), kmer, out.ctypes.data, cap,
                                            offs.ctypes.data, C.byref(used))
            if rc == -5:
                cap = used.value
                continue
            self._check(rc)
            break
        return [[tuple(int(x) for x in a) for a in out[offs[k]:offs[k + 1]]] for k in range(n)]

    def chain_batch(self, anchor_lists, max_chain_gap=210, match_chain_score=4):
        """GPU chain_anchors.  anchor_lists: one (m, 4) int32 array of (q, r, l, has_u) per pair.  Returns one
        (path, boundaries) per pair like the reference's chain_anchors (include/sedef_hip.h: sdf_chain_batch)."""
        n = len(anchor_lists)
        arrs = [np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 4) for a in anchor_lists]
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(a) for a in arrs])
        total = int(off[n])
        flat = np.concatenate(arrs) if total else np.zeros((0, 4), np.int32)
        path = np.zeros(max(total, 1), np.int32)
        bounds = np.zeros(2 * (total + n) + 2, np.int32)
        nb = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.sdf_chain_batch(self.ctx, flat.ctypes.data, off.ctypes.data, n, max_chain_gap,
                                             match_chain_score, path.ctypes.data, bounds.ctypes.data, nb.ctypes.data))
        out = []
        for i in range(n):
            b0 = 2 * (int(off[i]) + i)
            out.append((path[off[i]:off[i + 1]].copy(), bounds[b0:b0 + 2 * int(nb[i])].reshape(-1, 2).copy()))
        return out

    def stats_columns_batch(self, alignments):
        """Per-alignment columns of `stats generate`.  alignments: list of (a, b, cigar) with a, b the FASTA characters
        (str / bytes) and cigar uint32 runs len << 4 | op (0 'M', 1 'D', 2 'I').  Returns a STATS_COLS_DTYPE array
        (include/sedef_hip.h: sdf_stats_columns_batch)."""
        n = len(alignments)
        tasks = np.zeros(n, STATS_TASK_DTYPE)
        chunks, cigs, off, coff = [], [], 0, 0
        for k, (a, b, cg) in enumerate(alignments):
            a = a.encode() if isinstance(a, str) else bytes(a)
            b = b.encode() if isinstance(b, str) else bytes(b)
            cg = np.ascontiguousarray(cg, dtype=np.uint32)
            tasks[k] = (off, off + len(a), len(a), len(b), coff, len(cg), 0)
            off += len(a) + len(b)
            coff += len(cg)
            chunks += [a, b]
            cigs.append(cg)
        pool = b"".join(chunks)
        cig = np.concatenate(cigs) if cigs else np.zeros(0, np.uint32)
        out = np.zeros(n, STATS_COLS_DTYPE)
        self._check(self.lib.sdf_stats_columns_batch(self.ctx, tasks.ctypes.data, n, pool, len(pool), cig.ctypes.data,
                                                     len(cig), out.ctypes.data))
        return out

    def stats_columns_device(self, d_tasks, n, d_pool, d_cigar, d_out, stream=None):
        """The same over device pointers (ints); asynchronous on `stream` when one is given."""
        self._check(self.lib.sdf_stats_columns_device(self.ctx, d_tasks, n, d_pool, d_cigar, d_out, stream))

    def last_ms(self, which):
        return float(self.lib.sdf_last_ms(self.ctx, which))

    def last_launches(self):
        return int(self.lib.sdf_last_launches(self.ctx))

    def last_paired(self):
        """Tasks of the last batch that ran two per wavefront (same-geometry pairs)."""
        return int(self.lib.sdf_last_paired(self.ctx))

    def last_lane_tasks(self):
        """Tasks of the last batch that ran one per lane (extz2_lane.hip: small full-band tasks of a large batch)."""
        return int(self.lib.sdf_last_lane_tasks(self.ctx))

    def last_reran(self):
        """Tasks of the last batch that a stripe kernel gave up and the call ran again on another kernel."""
        return int(self.lib.sdf_last_reran(self.ctx))


_default_engine = None


def ksw_extz2(query, target, m=5, mat=None, q=40, e=1, w=-1, zdrop=-1, flag=0, engine=None):
    """Mirror of ksw_extz2_sse(km, qlen, query, tlen, target, m, mat, q, e, w, zdrop, flag, &ez)
    (reference: extern/ksw2.h:50).  Returns the ksw_extz_t fields as a dict + 'cigar' words."""
    global _default_engine
    if engine is None:
        if _default_engine is None:
            _default_engine = Extz2Engine()
        engine = _default_engine
    if m != 5:
        raise SdfError("GPU path implements m=5 only")
    want = WANT_ALL & ~WANT_CIGAR if flag & 1 else WANT_ALL
    res, cig = engine.align_pairs([(query, target)], w=w, zdrop=zdrop, flag=flag, mat=mat, gapo=q,
                                  gape=e, want=want)
    r = res[0]
    d = {k: int(r[k]) for k in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q",
                                "score")}
    d["cigar"] = cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])].copy()
    d["counts"] = {k: int(r[k]) for k in ("matches", "mismatches", "gaps", "gap_bases")}
    return d
