"""GPU fuzz campaign for the mixed pairs of round 4 (extz2_pair.hip, MIXED: two banded tasks of one band and different
lengths per wavefront) against the oracle; not collected by pytest, run by hand on a GPU box:
    SEED=1 ROUNDS=40 python tests/fuzz/fuzz_mixedpair.py
SDF_MIXED_MIN=2 is set here so that every chunk pairs what it can.  Score, mte, mte_q, zdropped, CIGAR (and the best cell of a
band that runs out) of every task must equal the oracle's; the first ten mismatching shapes are printed."""
import os, sys, time
os.environ.setdefault("SDF_MIXED_MIN", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sedef_amd
from oracle.binding import Oracle, mutate, random_codes, cigar_to_str
eng = sedef_amd.Extz2Engine(0)
orc = Oracle()
seed0 = int(os.environ.get("SEED", "1")); rounds = int(os.environ.get("ROUNDS", "20")); N = int(os.environ.get("N", "600"))
LMAX = int(os.environ.get("LMAX", "1800"))
bad = 0; total = 0; paired = 0; t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(seed0 * 104729 + rd)
    pairs, ws = [], []
    wset = [int(rng.integers(1, 560)) for _ in range(6)] + [15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 479, 480, 481, 511, 512, 513, 544]
    wset = [int(x) for x in rng.choice(wset, 8)]  # few bands per round: partners exist
    for _ in range(N):
        w = int(rng.choice(wset))
        ql = int(np.exp(rng.uniform(np.log(20), np.log(LMAX))))
        q = random_codes(rng, ql, 0.004 if rng.random() < 0.2 else 0.0)
        t = mutate(rng, q, float(rng.choice([0.0, 0.03, 0.1, 0.4])), 0.01, 0.01)
        if rng.random() < 0.3 and len(t) > 50:
            k, L = int(rng.integers(0, len(t) - 10)), int(rng.integers(1, w + 50))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else np.concatenate([t[:k], t[k + L:]])
        if len(t) == 0: t = random_codes(rng, 1)
        pairs.append((q, t)); ws.append(w)
    # a few score-only tasks and a few that leave their CIGAR reversed (KSW_EZ_REV_CIGAR): pairing classes of their own
    flags = [int(rng.choice([1, 0x80, 0x80])) if rng.random() < 0.12 else 0 for _ in range(N)]
    res, cig = eng.align_pairs(pairs, w=np.array(ws, np.int32), flag=np.array(flags, np.int32), want=3)
    paired += eng.last_paired()
    for (q, t), w, f, r in zip(pairs, ws, flags, res):
        exp = orc.extz2(q, t, w=w, flag=f)
        got = cigar_to_str(cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])])
        ok = got == cigar_to_str(exp["cigar"]) and all(int(r[f2]) == exp[f2] for f2 in ("score", "mte", "mte_q", "zdropped"))
        if exp["zdropped"]: ok = ok and (int(r["max_t"]), int(r["max_q"])) == (exp["max_t"], exp["max_q"])
        total += 1
        if not ok:
            bad += 1
            if bad <= 10: print("BAD", len(q), len(t), w, f, int(r["score"]), exp["score"], int(r["zdropped"]), exp["zdropped"], flush=True)
print("fuzz_mixedpair: %d tasks (%d ran two to a wavefront), %d bad, %.0f s" % (total, paired, bad, time.time() - t0))
