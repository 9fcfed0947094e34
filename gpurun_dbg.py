import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sedef_amd, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pool, q_off, qlen, t_off, tlen = bench.synth_batch(n, 1000, seed=42)
tasks = np.zeros(n, sedef_amd.TASK_DTYPE)
tasks["q_off"], tasks["t_off"], tasks["qlen"], tasks["tlen"] = q_off, t_off, qlen, tlen
tasks["w"], tasks["zdrop"] = 128, -1
eng = sedef_amd.Extz2Engine(0)
res, cig = eng.align_batch(tasks, pool, want=3)
bad = np.nonzero(res["score"] < -1000000)[0]
print("n", n, "bad", len(bad), "paired", eng.last_paired())
from collections import Counter
cnt = Counter(tlen.tolist())
for b in bad[:20]:
    print(b, "tlen", tlen[b], "count of this tlen", cnt[int(tlen[b])], {k: int(res[k][b]) for k in ("score", "mte", "mte_q", "n_cigar", "matches")})
odd = [k for k in range(n) if cnt[int(tlen[k])] % 2 == 1]
last_of = {}
for k in range(n): last_of[int(tlen[k])] = k
selfp = sorted(v for t, v in last_of.items() if cnt[t] % 2 == 1)
print("self-paired tasks", selfp)
print("ok self-pairs", [(k, int(tlen[k])) for k in selfp if k not in set(bad.tolist())])
print("bad", [(int(k), int(tlen[k])) for k in bad])
