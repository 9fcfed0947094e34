"""GPU fuzz campaign against the oracle (not collected by pytest: run by hand on a GPU box, e.g.
    SEED=1 ROUNDS=60 SDF_BSTRIPE_MIN_ROWS=100 SDF_BSTRIPE_ALL=1 [SDF_BSTRIPE_NREG=2] python tests/fuzz/fuzz_banded.py
Every task's score, mte, mte_q, zdropped, CIGAR (and the best cell of a band that runs out) must equal the oracle's; the
first ten mismatching shapes are printed.  Round 2 ran about 800,000 tasks through these three scripts with the stripe
kernels forced to every width; one parity bug came out of it (tests/golden/bstripe_refresh_spill.npz)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sedef_amd
from oracle.binding import Oracle, mutate, random_codes, cigar_to_str
eng = sedef_amd.Extz2Engine(0)
orc = Oracle()
seed0 = int(os.environ.get("SEED", "1")); rounds = int(os.environ.get("ROUNDS", "20")); N = 500
bad = 0; total = 0; t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(seed0 * 7919 + rd)
    pairs, ws = [], []
    for _ in range(N):
        w = int(rng.integers(1, 600)) if rng.random() < 0.7 else int(rng.choice([15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512]))
        ql = int(rng.integers(1, 1400))
        d = int(rng.integers(-w - 40, w + 41))
        tl = max(1, ql + d)
        q = random_codes(rng, ql, 0.004 if rng.random() < 0.2 else 0.0)
        t = mutate(rng, q, float(rng.choice([0.0, 0.03, 0.1, 0.4])), 0.01, 0.01)
        if rng.random() < 0.3 and len(t) > 50:
            k, L = int(rng.integers(0, len(t) - 10)), int(rng.integers(1, w + 50))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else np.concatenate([t[:k], t[k + L:]])
        if len(t) == 0: t = random_codes(rng, 1)
        t = t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])
        pairs.append((q, t)); ws.append(w)
    res, cig = eng.align_pairs(pairs, w=np.array(ws, np.int32), want=3)
    for (q, t), w, r in zip(pairs, ws, res):
        exp = orc.extz2(q, t, w=w)
        got = cigar_to_str(cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])])
        ok = got == cigar_to_str(exp["cigar"]) and all(int(r[f]) == exp[f] for f in ("score", "mte", "mte_q", "zdropped"))
        if exp["zdropped"]: ok = ok and (int(r["max_t"]), int(r["max_q"])) == (exp["max_t"], exp["max_q"])
        total += 1
        if not ok:
            bad += 1
            if bad <= 10: print("BAD", len(q), len(t), w, int(r["score"]), exp["score"], int(r["zdropped"]), exp["zdropped"], flush=True)
print("fuzz_band: %d tasks, %d bad, %.0f s" % (total, bad, time.time() - t0))
