"""Shared helpers for the test-suite (test infrastructure, may import oracle/)."""
import numpy as np

from oracle.binding import cigar_to_str, sedef_mat  # noqa: F401

ALPHA = "ACGTN"
_LUT = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGT"):
    _LUT[ord(_c)] = _i
    _LUT[ord(_c.lower())] = _i


def codes(s):
    return _LUT[np.frombuffer(s.encode(), dtype=np.uint8)]


FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score")


def check_case(got, case):
    exp = case["expect"]
    for k in FIELDS:
        assert got[k] == exp[k], "%s: field %s got %r want %r" % (case["tag"], k, got[k], exp[k])
    assert cigar_to_str(got["cigar"]) == exp["cigar"], case["tag"]
