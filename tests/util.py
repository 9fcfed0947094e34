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


def random_stats_case(rng, k):
    """One (a, b, runs) alignment for the stats-columns tests: runs = [(op, len)] with op 0 'M', 1 'D' (a only),
    2 'I' (b only).  Mixed case, N runs, zero-length runs, more than 64 / 128 runs, long runs, a CIGAR that stops
    short of the sequences, empty CIGARs -- chosen by k so that every generator run holds every kind."""
    kind = k % 8
    n_runs = [0, 1, 3, 40, 64, 65, 150, 400][kind]
    if k % 29 == 0:
        n_runs = int(rng.integers(0, 700))
    max_len = [1, 5000, 300, 60, 8, 130, 40, 25][kind]
    if n_runs * max_len > 40000:
        max_len = max(1, 40000 // n_runs)
    runs, prev = [], -1
    for _ in range(n_runs):
        op = int(rng.choice([0, 0, 0, 1, 2]))
        if op == prev and rng.random() < 0.7:
            op = 0 if op else int(rng.integers(1, 3))
        ln = 0 if rng.random() < 0.03 else int(rng.integers(1, max_len + 1))
        runs.append((op, ln))
        prev = op
    na = sum(l for op, l in runs if op != 2) + (int(rng.integers(0, 20)) if k % 3 == 0 else 0)
    nb = sum(l for op, l in runs if op != 1) + (int(rng.integers(0, 20)) if k % 5 == 0 else 0)

    def seq(n, base):
        s = base[:n].copy() if base is not None and len(base) >= n else rng.choice(list(b"ACGT"), n).astype(np.uint8)
        if base is not None and len(base) >= n:
            mut = rng.random(n) < 0.12
            s[mut] = rng.choice(list(b"ACGT"), int(mut.sum()))
        else:
            s = np.asarray(s, np.uint8)
        lo, pos = rng.random() < 0.5, 0
        while pos < n:  # soft-masked stretches
            step = int(rng.integers(1, 200))
            if lo:
                s[pos:pos + step] |= 0x20
            lo, pos = not lo, pos + step
        for _ in range(int(rng.integers(0, 3))):  # N runs, either case
            p, ln = int(rng.integers(0, n + 1)), int(rng.integers(1, 40))
            s[p:p + ln] = ord("N") if rng.random() < 0.6 else ord("n")
        return s

    a = seq(na, None)
    # b follows a along the M runs, so that matches dominate like in a real alignment
    b_src = np.zeros(nb, np.uint8)
    ia = ib = 0
    for op, ln in runs:
        if op == 0:
            b_src[ib:ib + ln] = a[ia:ia + ln] & 0xDF
        elif op == 2:
            b_src[ib:ib + ln] = rng.choice(list(b"ACGT"), ln)
        ia += ln if op != 2 else 0
        ib += ln if op != 1 else 0
    b_src[ib:] = rng.choice(list(b"ACGT"), nb - ib)
    b_src[b_src == ord("N") & 0xDF] = ord("N")
    b = seq(nb, b_src)
    return a.tobytes().decode(), b.tobytes().decode(), runs
