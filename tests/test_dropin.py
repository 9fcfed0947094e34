"""The exact drop-in entry sdf_ksw_extz2 (include/sedef_hip.h <-> reference extern/ksw2.h:22-30,50), align_helper's
60 kb chunk loop (src/align.cc:46-57) and the CLI's scoring overrides (src/align_main.cc:343-352)."""
import ctypes as C
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hostgen
from oracle.binding import _KswExtz  # ksw_extz_t, field for field
from util import cigar_to_str, codes, sedef_mat

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


def _ref_hook():
    from oracle.binding import build_reference
    so = build_reference()
    if not so:
        return None, None
    lib = C.CDLL(so)
    return (C.cast(lib.ref_extz2_hook, C.c_void_p), lib) if hasattr(lib, "ref_extz2_hook") else (None, None)


# ---------------------------------------------------------------- G1: scoring overrides, pinned by the reference class
def test_alignment_scoring_overrides_match_reference_golden(host, oracle, host_golden):
    hook = C.cast(oracle.lib.sdfo_extz2, C.c_void_p)
    assert len(host_golden["scored"]) >= 30
    for c in host_golden["scored"]:
        cig, cnt = host.alignment_pair(c["a"], c["b"], test_dp=hook, scoring=c["scoring"])
        assert cig == c["cigar"] and cnt == c["counts"], c["scoring"]
    c = host_golden["pairs"][0]  # the override does not leak into later default-scoring calls
    assert host.alignment_pair(c["a"], c["b"], test_dp=hook) == (c["cigar"], c["counts"])


# ---------------------------------------------------------------- A1: the 60 kb chunk loop
def _check_chunked(host, case, test_dp):
    a, b = hostgen.chunk_case(case["seed"], case["n"], case["d"])
    assert (len(a), len(b)) == (case["len_a"], case["len_b"])
    cig, cnt = host.alignment_pair(a, b, test_dp=test_dp)
    assert cnt == case["counts"] and len(cig) == case["cigar_len"]
    assert cig[:60] == case["cigar_head"] and cig[-60:] == case["cigar_tail"]
    assert hashlib.sha256(cig.encode()).hexdigest() == case["cigar_sha256"]


def test_chunk_loop_matches_reference_golden_cpu(host, host_golden):
    """Two sequences of > 60,000 bases: two ksw calls at equal offsets, CIGARs concatenated (src/align.cc:46-57).
    Expected values come from the reference's own Alignment class; the DP here is the reference kernel."""
    hook, _keep = _ref_hook()
    if hook is None:
        pytest.skip("oracle/_ref not built (the scalar oracle would need minutes for 3.6e9 cells)")
    _check_chunked(host, host_golden["chunked"][1], hook)


@pytest.mark.gpu
def test_chunk_loop_matches_reference_golden_gpu(host, host_golden):
    for case in host_golden["chunked"]:
        _check_chunked(host, case, None)


# ---------------------------------------------------------------- (b) boundary: sdf_ksw_extz2 itself
UNSUPPORTED = 0  # (every KSW_EZ_* flag of the extz2 kernel is served)

def _call_dropin(lib, libc, q, t, mat, gapo, gape, w, zdrop, flag):
    ez = _KswExtz()
    # poison: the callee must overwrite every field (ksw_reset_extz, extern/ksw2.h:153-159)
    C.memset(C.byref(ez), 0x5A, C.sizeof(ez))
    qa = np.ascontiguousarray(q, np.uint8)
    ta = np.ascontiguousarray(t, np.uint8)
    m = np.ascontiguousarray(mat, np.int8)
    lib.sdf_ksw_extz2(None, len(qa), qa.ctypes.data_as(C.POINTER(C.c_uint8)), len(ta),
                      ta.ctypes.data_as(C.POINTER(C.c_uint8)), 5, m.ctypes.data_as(C.POINTER(C.c_int8)), gapo, gape, w,
                      zdrop, flag, C.byref(ez))
    cig = np.ctypeslib.as_array(ez.cigar, shape=(ez.n_cigar,)).copy() if ez.n_cigar else np.zeros(0, np.uint32)
    got = dict(max=ez.max_zd & 0x7fffffff, zdropped=ez.max_zd >> 31, max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe,
               mqe_t=ez.mqe_t, mte=ez.mte, mte_q=ez.mte_q, score=ez.score, cigar=cig, m_cigar=ez.m_cigar,
               n_cigar=ez.n_cigar, cigar_ptr=bool(ez.cigar))
    if ez.cigar:
        libc.free(C.cast(ez.cigar, C.c_void_p))  # caller frees, like src/align.cc:65
    return got


@pytest.mark.gpu
def test_sdf_ksw_extz2_dropin_on_golden_vectors(golden_cases):
    import sedef_amd
    lib = sedef_amd.load_library()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    lib.sdf_ksw_extz2.restype = None
    lib.sdf_ksw_extz2.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_int8,
                                  C.POINTER(C.c_int8), C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(_KswExtz)]
    assert C.sizeof(_KswExtz) == 56  # extern/ksw2.h:22-30 on LP64
    n = 0
    for c in golden_cases:
        if c["flag"] & UNSUPPORTED:
            continue
        got = _call_dropin(lib, libc, codes(c["q"]), codes(c["t"]), sedef_mat(c["match"], c["mismatch"]), c["gapo"],
                           c["gape"], c["w"], c["zdrop"], c["flag"])
        exp = c["expect"]
        for k in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score"):
            assert got[k] == exp[k], (c["tag"], k, got[k], exp[k])
        assert cigar_to_str(got["cigar"]) == exp["cigar"], c["tag"]
        assert got["n_cigar"] == len(got["cigar"]) and got["m_cigar"] >= got["n_cigar"]
        assert got["cigar_ptr"] == (got["n_cigar"] > 0)  # ez->cigar = 0 in the reset (extern/ksw2.h:158)
        n += 1
    assert n == len(golden_cases) >= 284
    # empty inputs: the reset result, no CIGAR (extern/ksw2_extz2_sse.cc:57)
    for (q, t) in ((np.zeros(0, np.uint8), codes("ACGT")), (codes("ACGT"), np.zeros(0, np.uint8))):
        got = _call_dropin(lib, libc, q, t, sedef_mat(), 40, 1, -1, -1, 0)
        assert got["score"] == -0x40000000 and got["n_cigar"] == 0 and not got["cigar_ptr"]
        assert (got["max"], got["max_q"], got["max_t"], got["mqe_t"], got["mte_q"]) == (0, -1, -1, -1, -1)


@pytest.mark.gpu
def test_sdf_ksw_extz2_fatal_error_exits_120():
    """A request the GPU path cannot serve ends the process with 120 like the reference's own failure mode
    (extern/ksw2.h:106), never with a wrong answer."""
    code = ("import ctypes as C, numpy as np, sys; sys.path.insert(0, %r); import sedef_amd; "
            "from oracle.binding import _KswExtz; lib = sedef_amd.load_library(); ez = _KswExtz(); "
            "q = np.zeros(40, np.uint8); mat = np.zeros(36, np.int8); "
            "lib.sdf_ksw_extz2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, "
            "C.c_int8, C.c_int8, C.c_int, C.c_int, C.c_int, C.c_void_p]; "
            "lib.sdf_ksw_extz2(None, 40, q.ctypes.data, 40, q.ctypes.data, 6, mat.ctypes.data, 40, 1, -1, -1, 0, "
            "C.addressof(ez))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 120 and "sdf_ksw_extz2" in r.stderr
