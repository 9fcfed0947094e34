"""CPU: the oracle restatement reproduces every golden vector produced by the reference kernel."""
from util import check_case, codes, sedef_mat


def test_oracle_matches_reference_golden_vectors(oracle, golden_cases):
    assert len(golden_cases) >= 250
    for c in golden_cases:
        got = oracle.extz2(codes(c["q"]), codes(c["t"]), mat=sedef_mat(c["match"], c["mismatch"]),
                           gapo=c["gapo"], gape=c["gape"], w=c["w"], zdrop=c["zdrop"],
                           flag=c["flag"])
        check_case(got, c)


def test_band_cells_formula(oracle):
    # 1000x1000 cell counts per band
    assert oracle.band_cells(1000, 1000, 64) == 124840
    assert oracle.band_cells(1000, 1000, 128) == 240488
    assert oracle.band_cells(1000, 1000, 512) == 762344
    assert oracle.band_cells(1000, 1000, -1) == 1000000
    assert oracle.band_cells(7, 1000, -1) == 7000


def test_counts_restatement(oracle):
    import numpy as np
    q = np.array([0, 1, 2, 3, 4, 0, 0], np.uint8)
    t = np.array([0, 1, 3, 3, 4, 0], np.uint8)
    cig = np.array([5 << 4 | 0, 1 << 4 | 1, 1 << 4 | 0], np.uint32)  # 5M1I1M
    c = oracle.counts(cig, q, t)
    assert c == dict(matches=4, mismatches=2, gaps=1, gap_bases=1)
