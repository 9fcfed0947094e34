"""ctypes access to the host pipeline library (sedef_amd/lib/libsedef_host.so, sources in csrc/host/).

`test_dp` parameters are the library's TEST HOOK: a C function with the oracle's single-task signature that
replaces the GPU provider.  Product code (the `sedef` CLI) never passes one."""
import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SRC = os.path.join(_HERE, "csrc", "host")
LIB = os.path.join(_HERE, "lib", "libsedef_host.so")
CLI = os.path.join(_HERE, "bin", "sedef")
MULTI = os.path.join(_HERE, "bin", "sdf_multi")
_SOURCES = ["alignment.cc", "hit_fasta.cc", "chain.cc", "pipeline.cc", "bucket.cc", "stats.cc", "host_cabi.cc"]


def build_host(force=False):
    """g++ build of the host library and the `sedef` CLI (needs libsedef_hip.so next to it)."""
    srcs = [os.path.join(HOST_SRC, f) for f in _SOURCES]
    deps = srcs + [os.path.join(HOST_SRC, "sedef_host.h"), os.path.join(HOST_SRC, "sedef_main.cc"),
                   os.path.join(HOST_SRC, "multi_gpu.cc")]
    multi_ok = os.path.exists(MULTI) or os.path.exists(MULTI + ".skipped")
    built = [LIB, CLI] + ([MULTI] if os.path.exists(MULTI) else [])
    fresh = os.path.exists(LIB) and os.path.exists(CLI) and multi_ok and all(
        os.path.getmtime(d) <= min(os.path.getmtime(b) for b in built) for d in deps)
    if fresh and not force:
        return LIB
    gxx = shutil.which("g++") or "g++"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    libdir = os.path.join(_HERE, "lib")
    subprocess.check_call([gxx, "-O2", "-g", "-std=c++17", "-fPIC", "-Wall", "-shared", "-o", LIB] + srcs +
                          ["-L" + libdir, "-lsedef_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    subprocess.check_call([gxx, "-O2", "-g", "-rdynamic", "-std=c++17", "-Wall", "-o", CLI, os.path.join(HOST_SRC, "sedef_main.cc"),
                           "-L" + libdir, "-lsedef_host", "-lsedef_hip", "-lpthread", "-Wl,-rpath,$ORIGIN/../lib"])
    # sdf_multi: one batch sharded over the GPUs of a node + RCCL all-gatherv of the results, host C++ above the C ABI.  The
    # file holds no kernels: g++ against the HIP runtime headers; a box without them (CPU-only use of the host library and
    # its test provider) skips the tool instead of failing the build.
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    try:
        subprocess.check_call([gxx, "-O2", "-std=c++17", "-Wall", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__",
                               "-I" + os.path.join(rocm, "include"), "-o", MULTI, os.path.join(HOST_SRC, "multi_gpu.cc"),
                               "-L" + libdir, "-lsedef_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lpthread",
                               "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.join(rocm, "lib")])
        if os.path.exists(MULTI + ".skipped"):
            os.remove(MULTI + ".skipped")
    except (subprocess.CalledProcessError, OSError) as e:
        print("sedef_amd.host: sdf_multi not built (%s); the library and the CLI are complete without it" % e)
        if not os.path.exists(MULTI):
            open(MULTI + ".skipped", "w").write(str(e))
    return LIB


_lib = None


def load_host():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB)
        _lib = C.CDLL(LIB)
        _lib.sdfh_last_error.restype = C.c_char_p
    return _lib


def _err(lib, rc):
    if rc != 0:
        raise RuntimeError("host pipeline error %d: %s" % (rc, lib.sdfh_last_error().decode()))


def generate(ref_path, bed_path, kmer, out_path, match=5, mismatch=-4, gap_open=-40, gap_extend=-1, test_dp=None,
             device=0):
    """`sedef align generate` (reference: src/align_main.cc:285-337). Returns (lines, hits, dp_tasks, dp_cells, rounds)."""
    lib = load_host()
    stats = (C.c_longlong * 5)()
    rc = lib.sdfh_generate(ref_path.encode(), bed_path.encode(), kmer, out_path.encode(), match, mismatch, gap_open,
                           gap_extend, test_dp, device, stats)
    _err(lib, rc)
    return tuple(int(x) for x in stats)


def generate_many(ref_path, beds, kmer, out_suffix=".aligned.bed", log_dir=None, test_dp=None, device=0):
    """Several buckets through ONE provider (csrc/host/pipeline.cc: generate_many): bucket b's lines go to b + out_suffix.
    `beds`: bucket files or directories of bucket_???? files.  Returns [(lines, hits)] per bucket."""
    lib = load_host()
    # (two figures per bucket; a directory stands for the bucket_???? files in it)
    nb = sum(len(os.listdir(b)) if os.path.isdir(b) else 1 for b in beds)
    stats = (C.c_longlong * (2 * max(nb, 1) + 2))()
    lib.sdfh_generate_many.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
    n = lib.sdfh_generate_many(ref_path.encode(), "\n".join(beds).encode(), kmer, out_suffix.encode(),
                               log_dir.encode() if log_dir else None, test_dp, device, stats)
    _err(lib, min(n, 0))
    return [(int(stats[2 * k]), int(stats[2 * k + 1])) for k in range(n)]


_buf = None


def _buffer():
    global _buf
    if _buf is None:
        _buf = C.create_string_buffer(4 << 20)
    return _buf


def alignment_pair(fa, fb, test_dp=None, device=0, scoring=None):
    """Alignment(fa, fb) (reference: src/align.cc:76-88) -> (CIGAR string, [matches, mismatches, gaps, gap_bases, span]).
    scoring = (match, mismatch, gap_open, gap_extend) as the CLI overrides (src/align_main.cc:343-352)."""
    lib, buf = load_host(), _buffer()
    cnt = (C.c_int * 5)()
    if scoring is None:
        _err(lib, lib.sdfh_alignment_pair(fa.encode(), fb.encode(), test_dp, device, buf, len(buf), cnt))
    else:
        _err(lib, lib.sdfh_alignment_pair_scored(fa.encode(), fb.encode(), *[int(x) for x in scoring], test_dp, device,
                                                 buf, len(buf), cnt))
    return buf.value.decode(), list(cnt)


def guide_from_chains(q, r, spec, side, test_dp=None, device=0):
    lib, buf = load_host(), _buffer()
    _err(lib, lib.sdfh_guide_from_chains(q.encode(), r.encode(), spec.encode(), side, test_dp, device, buf, len(buf)))
    return buf.value.decode()


def fast_align(query, ref, qname="q", rname="r", q_rc=False, r_rc=False, qstart=0, rstart=0, kmer=11, test_dp=None,
               device=0):
    """fast_align (reference: src/chain.cc:203-268) -> [(qs, qe, rs, re, cigar, matches, mismatches, gaps, gap_bases)]."""
    lib, buf = load_host(), _buffer()
    lib.sdfh_fast_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    _err(lib, lib.sdfh_fast_align(query.encode(), ref.encode(), qname.encode(), rname.encode(), int(q_rc), int(r_rc),
                                  qstart, rstart, kmer, test_dp, device, buf, len(buf)))
    out = []
    for line in buf.value.decode().split("\n"):
        if line:
            f = line.split(" ")
            out.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4]) + tuple(int(x) for x in f[5:9]))
    return out


def chains(q, r, kmer=11):
    lib, buf = load_host(), _buffer()
    lib.sdfh_chains.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    _err(lib, lib.sdfh_chains(q.encode(), r.encode(), kmer, buf, len(buf)))
    out = []
    for c in buf.value.decode().split("|"):
        if c:
            out.append([tuple(int(x) for x in a.split()) for a in c.split(";") if a])
    return out


def chain_raw(anchors, max_chain_gap=210, match_chain_score=4):
    """chain_anchors on an (m, 4) int32 array of (q, r, l, has_u): returns (path, boundaries)."""
    import numpy as np
    lib = load_host()
    a = np.ascontiguousarray(anchors, dtype=np.int32).reshape(-1, 4)
    m = len(a)
    path = np.zeros(max(m, 1), np.int32)
    bounds = np.zeros(2 * (m + 1), np.int32)
    lib.sdfh_chain_raw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    nb = lib.sdfh_chain_raw(a.ctypes.data, m, max_chain_gap, match_chain_score, path.ctypes.data, bounds.ctypes.data)
    _err(lib, min(nb, 0))
    return path[:m].copy(), bounds[:2 * nb].reshape(-1, 2).copy()


def rangemax_script(pts, ops):
    """Test hook: activate / deactivate / range-maximum calls on the host's priority search tree (chain.cc: RangeMax), in the
    script format of oracle/ref_align_driver.cc: ref_segtree_script.  Returns (out [k, 2], top pointer of every node)."""
    import numpy as np
    lib = load_host()
    pts = np.ascontiguousarray(pts, dtype=np.int32).reshape(-1, 2)
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 5)
    out = np.zeros((max(len(ops), 1), 2), np.int32)
    cap = 4 << max(1, int(len(pts) - 1).bit_length())
    state = np.zeros(cap, np.int32)
    lib.sdfh_rangemax_script.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    size = lib.sdfh_rangemax_script(pts.ctypes.data, len(pts), ops.ctypes.data, len(ops), out.ctypes.data, state.ctypes.data, cap)
    return out[:len(ops)].copy(), state[:size].copy()


def fasta_get(path, name, start, end):
    lib, buf = load_host(), _buffer()
    e = C.c_int(0 if end is None else end)
    _err(lib, lib.sdfh_fasta_get(path.encode(), name.encode(), start, None if end is None else C.byref(e), buf,
                                 len(buf)))
    return buf.value.decode(), (None if end is None else e.value)


def merge(bed_lines, merge_dist=250):
    lib, buf = load_host(), _buffer()
    _err(lib, lib.sdfh_merge("\n".join(bed_lines).encode(), merge_dist, buf, len(buf)))
    return buf.value.decode()


def stats_generate(ref_path, bed_path, out_path, max_ok_gap=-1, min_split=1000, uppercase=100, max_error=0.5,
                   test_cols=None, device=0):
    """`sedef stats generate` (reference: src/stats_main.cc:339-389): the table of the final calls.  test_cols: a column
    walker with the oracle's signature (oracle/stats_oracle.c) instead of the device.  Returns (lines, hits, pieces, columns)."""
    lib = load_host()
    st = (C.c_longlong * 3)()
    lib.sdfh_stats_generate.restype = C.c_long
    lib.sdfh_stats_generate.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                        C.c_int, C.c_void_p]
    n = lib.sdfh_stats_generate(ref_path.encode(), bed_path.encode(), out_path.encode(), max_ok_gap, min_split, uppercase,
                                max_error, test_cols, device, st)
    _err(lib, min(n, 0))
    return (int(n),) + tuple(int(x) for x in st)


def format_double(x):
    lib, buf = load_host(), _buffer()
    lib.sdfh_format_double.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    _err(lib, lib.sdfh_format_double(float(x), buf, len(buf)))
    return buf.value.decode()


def bucket(bed_path, nbins, out_dir, reference):
    lib = load_host()
    _err(lib, lib.sdfh_bucket(bed_path.encode(), nbins, out_dir.encode(), reference.encode()))


def anchors(q, r, kmer=11, same_chr=False, qstart=0, rstart=0):
    """Host generate_anchors (reference: src/chain.cc:24-101) -> list of (q, r, l, has_u)."""
    lib, buf = load_host(), _buffer()
    lib.sdfh_anchors.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    _err(lib, lib.sdfh_anchors(q.encode(), r.encode(), kmer, int(same_chr), qstart, rstart, buf, len(buf)))
    return [tuple(int(x) for x in a.split()) for a in buf.value.decode().split(";") if a]


def hit_extend(qs, qe, rs, re_, factor=5.0, max_extend=15000):
    """Hit::extend (reference: src/hit.cc:200-207)."""
    lib = load_host()
    io = (C.c_int * 4)(qs, qe, rs, re_)
    lib.sdfh_hit_extend.argtypes = [C.c_void_p, C.c_double, C.c_int]
    _err(lib, lib.sdfh_hit_extend(io, factor, max_extend))
    return list(io)


def sequence(name, seq, is_rc=False):
    """Sequence ctor (reference: src/hash.cc:104-109) -> (name, seq, is_rc)."""
    lib, buf = load_host(), _buffer()
    _err(lib, lib.sdfh_sequence(name.encode(), seq.encode(), int(is_rc), buf, len(buf)))
    n, s, r = buf.value.decode().split("|")
    return n, s, r == "1"
