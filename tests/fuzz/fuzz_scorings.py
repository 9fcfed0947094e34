"""GPU fuzz against the oracle over SCORINGS (not collected by pytest: run by hand on a GPU box, e.g.
    SEED=1 ROUNDS=40 python tests/fuzz/fuzz_scorings.py
Round 5: the register-resident kernels take three differences of the recurrence with 32-bit subtracts, which is exact when both
fresh score bytes match + 2 (q + e) and mismatch + 2 (q + e) lie in q .. 127 (sedef_amd/csrc/sdf_api.hip: scoring_gates);
every other scoring runs on the general kernel.  Each round draws ONE scoring -- a third of them on the edges of that
condition (mismatch + q + 2 e = 0 / -1, match + 2 (q + e) = 127 / 128, q = 0, e = 0), the rest anywhere the reference accepts
-- and a batch of banded and full-band tasks of all shapes; every task's score, mte, mte_q, zdropped and CIGAR (and the best
cell of a band that runs out) must equal the oracle's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sedef_amd
from sedef_amd.extz2 import sedef_mat
from oracle.binding import Oracle, mutate, random_codes, cigar_to_str
eng = sedef_amd.Extz2Engine(0)
orc = Oracle()
seed0 = int(os.environ.get("SEED", "1")); rounds = int(os.environ.get("ROUNDS", "20")); N = int(os.environ.get("N", "300"))
maxlen = int(os.environ.get("MAXLEN", "1600"))
bad = 0; total = 0; fast_rounds = 0; t0 = time.time()
for rd in range(rounds):
    rng = np.random.default_rng(seed0 * 4241 + rd)
    kind = int(rng.integers(0, 9))
    go, ge = int(rng.integers(0, 62)), int(rng.integers(0, 6))
    ma, mi = int(rng.integers(0, 14)), -int(rng.integers(1, 16))
    if kind == 0: mi = -(go + 2 * ge) if go + 2 * ge > 0 else -1          # mismatch + q + 2 e = 0: the last fast one
    if kind == 1: mi = -(go + 2 * ge) - 1                                   # ... and the first on the general kernel
    if kind == 2: ma = 127 - 2 * (go + ge)                                  # cap = 127
    if kind == 3: ma = 128 - 2 * (go + ge)                                  # cap = 128: the bytes wrap
    if kind == 2 or kind == 3:
        if ma < 0 or ma > 127: go, ge, ma = 40, 1, (45 if kind == 2 else 46)  # (an int8 matrix entry)
    if -mi > 2 * (go + ge) and rng.random() < 0.8: mi = -min(-mi, max(1, 2 * (go + ge)))  # (else the reference returns at once)
    qe2 = 2 * (go + ge)
    fast = ma + qe2 <= 127 and mi + qe2 <= 127 and ma + qe2 >= go and mi + qe2 >= go
    fast_rounds += fast
    mat = sedef_mat(ma, mi)
    pairs, ws = [], []
    for _ in range(N):
        ql = int(np.exp(rng.uniform(np.log(1), np.log(maxlen))))
        tl = max(1, ql + int(rng.integers(-200, 200)))
        w = int(rng.choice([-1, -1, 1, 3, 15, 16, 17, 33, 64, 100, 128, 129, 255, 256, 400, 512]))
        q = random_codes(rng, ql, 0.004 if rng.random() < 0.3 else 0.0)
        t = mutate(rng, q, float(rng.choice([0.0, 0.03, 0.1, 0.4])), 0.01, 0.01) if rng.random() < 0.85 else random_codes(rng, tl)
        if rng.random() < 0.3 and len(t) > 50:
            k, L = int(rng.integers(0, len(t) - 10)), int(rng.integers(1, 400))
            t = np.concatenate([t[:k], random_codes(rng, L), t[k:]]) if rng.random() < 0.5 else np.concatenate([t[:k], t[k + L:]])
        if len(t) == 0: t = random_codes(rng, 1)
        t = t[:tl] if len(t) >= tl else np.concatenate([t, random_codes(rng, tl - len(t))])
        pairs.append((q, t)); ws.append(w)
    res, cig = eng.align_pairs(pairs, w=np.array(ws, np.int32), want=3, mat=mat, gapo=go, gape=ge)
    for (q, t), w, r in zip(pairs, ws, res):
        exp = orc.extz2(q, t, w=w, mat=mat, gapo=go, gape=ge)
        got = cigar_to_str(cig[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])])
        ok = got == cigar_to_str(exp["cigar"]) and all(int(r[f]) == exp[f] for f in ("score", "mte", "mte_q", "zdropped"))
        if exp["zdropped"]: ok = ok and (int(r["max_t"]), int(r["max_q"])) == (exp["max_t"], exp["max_q"])
        total += 1
        if not ok:
            bad += 1
            if bad <= 10: print("BAD", (ma, mi, go, ge), len(q), len(t), w, int(r["score"]), exp["score"], flush=True)
print("fuzz_scorings: %d tasks under %d scorings (%d on the register-resident kernels), %d bad, %.0f s" % (total, rounds, fast_rounds, bad, time.time() - t0))
