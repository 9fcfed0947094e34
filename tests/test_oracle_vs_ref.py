"""CPU: live fuzz of the oracle against the reference kernel compiled from /root/reference.

Skipped where oracle/_ref/libksw2_ref.so is absent and cannot be built (no reference checkout).
"""
import numpy as np
import pytest

from oracle.binding import Reference, mutate, random_codes, sedef_mat
from util import FIELDS


@pytest.fixture(scope="module")
def ref():
    if not Reference.available():
        pytest.skip("reference kernel not built (oracle/_ref) and /root/reference absent")
    return Reference()


def _same(a, b):
    return all(a[k] == b[k] for k in FIELDS) and np.array_equal(a["cigar"], b["cigar"])


def test_fuzz_sedef_scoring(oracle, ref):
    rng = np.random.default_rng(11)
    for _ in range(3000):
        q = random_codes(rng, int(rng.integers(1, 400)), 0.02 if rng.random() < 0.3 else 0.0)
        d = rng.random() * 0.12
        t = mutate(rng, q, d, d / 3, d / 3)
        if rng.random() < 0.3:
            k, L = int(rng.integers(0, len(t))), int(rng.integers(1, 80))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]])
        kw = dict(w=int(rng.choice([-1, 0, 1, 2, 7, 15, 16, 17, 31, 32, 33, 64, 128])),
                  flag=int(rng.choice([0, 0, 0, 2, 0x40, 0x80, 1, 8, 4, 0x18])),
                  zdrop=int(rng.choice([-1, -1, 50, 200])))
        assert _same(oracle.extz2(q, t, **kw), ref.extz2(q, t, **kw)), kw


def test_fuzz_wrapping_scorings(oracle, ref):
    rng = np.random.default_rng(12)
    for _ in range(3000):
        q = random_codes(rng, int(rng.integers(1, 300)), 0.05 if rng.random() < 0.3 else 0.0)
        d = rng.random() * 0.2
        t = mutate(rng, q, d, d / 3, d / 3)
        kw = dict(w=int(rng.choice([-1, 0, 1, 3, 8, 15, 16, 17, 40, 100])),
                  mat=sedef_mat(int(rng.integers(1, 12)), -int(rng.integers(1, 12))),
                  gapo=int(rng.integers(0, 70)), gape=int(rng.integers(0, 6)),
                  flag=int(rng.choice([0, 0, 2, 0x40])))
        assert _same(oracle.extz2(q, t, **kw), ref.extz2(q, t, **kw)), kw


def test_fuzz_generic_matrix_and_approximate_modes(oracle, ref):
    """KSW_EZ_GENERIC_SC with a matrix whose wildcard row / column is not zero, KSW_EZ_APPROX_MAX / APPROX_DROP."""
    rng = np.random.default_rng(13)
    mat = np.array([6, -3, -5, -3, -1, -3, 6, -3, -5, -1, -5, -3, 6, -3, -1, -3, -5, -3, 6, -1, -1, -1, -1, -1, 1], np.int8)
    for _ in range(2000):
        q = random_codes(rng, int(rng.integers(1, 300)), 0.04)
        d = rng.random() * 0.2
        t = mutate(rng, q, d, d / 3, d / 3)
        flag = int(rng.choice([0x04, 0x08, 0x18, 0x0c, 0x1c, 0x48, 0x09, 0x05]))
        kw = dict(w=int(rng.choice([-1, -1, 3, 17, 64])), zdrop=int(rng.choice([-1, 40, 300])), flag=flag,
                  mat=mat if flag & 4 else sedef_mat(), gapo=12, gape=2)
        assert _same(oracle.extz2(q, t, **kw), ref.extz2(q, t, **kw)), kw
