"""GPU fuzz of the stats-columns kernel against oracle/stats_oracle.c (run by hand on a GPU box):
    SEED=1 ROUNDS=20 python tests/fuzz/fuzz_stats_columns.py
Random alignments of every kind tests/util.py:random_stats_case makes, plus alphabets that stress the packed predicates:
every ASCII byte, bytes above 0x7f, letters next to the range ends ('@', '[', '`', '{'), dashes inside the sequences."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sedef_amd
from oracle.binding import Oracle, STATS_FIELDS
from util import random_stats_case
eng = sedef_amd.Extz2Engine(0)
orc = Oracle()
seed0 = int(os.environ.get("SEED", "1")); rounds = int(os.environ.get("ROUNDS", "10")); N = int(os.environ.get("N", "1500"))
ALPHABETS = [b"ACGT", b"ACGTacgtNn", b"ACGTacgtNn-", bytes(range(1, 128)), bytes(range(1, 256)), b"@AZ[`az{GgCcTtNn-MRYK"]
bad = total = 0; t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(seed0 * 1000 + rd)
    cases = []
    for k in range(N):
        a, b, runs = random_stats_case(rng, k + rd)
        if k % 3 == 0:  # re-letter the sequences
            al = np.frombuffer(ALPHABETS[int(rng.integers(len(ALPHABETS)))], np.uint8)
            a = rng.choice(al, len(a)).astype(np.uint8).tobytes()
            b2 = np.frombuffer(a, np.uint8)[:len(b)].copy() if len(a) >= len(b) and rng.random() < 0.5 else rng.choice(al, len(b)).astype(np.uint8)
            mut = rng.random(len(b2)) < 0.2
            b2[mut] = rng.choice(al, int(mut.sum()))
            b = b2.tobytes()
        cases.append((a, b, np.array([(l << 4) | op for op, l in runs], np.uint32)))
    got = eng.stats_columns_batch(cases)
    for k, (a, b, cg) in enumerate(cases):
        exp = orc.stats_columns(a, b, cg)
        if [int(got[k][f]) for f in STATS_FIELDS] != exp.tolist():
            bad += 1
            if bad <= 10:
                print("MISMATCH round", rd, "case", k, len(a), len(b), len(cg), [int(got[k][f]) for f in STATS_FIELDS], exp.tolist())
    total += len(cases)
print("fuzz_stats: %d alignments, %d bad, %.0f s" % (total, bad, time.time() - t0))
