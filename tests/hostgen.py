"""Seeded synthetic inputs for the host-pipeline tests (test infrastructure)."""
import numpy as np

AL = np.array(list("ACGTacgtN"))


def rseq(rng, n, pn=0.0):
    p = np.array([.2, .2, .2, .2, .05, .05, .05, .05, 0.0])
    if pn:
        p = p * (1 - pn)
        p[8] = pn
    return "".join(rng.choice(AL, n, p=p / p.sum()))


def mut(rng, s, d):
    out = []
    for c in s:
        x = rng.random()
        if x < d * 0.6:
            out.append(str(rng.choice(list("ACGTacgt"))))
        elif x < d * 0.8:
            continue
        elif x < d:
            out.append(c)
            out.append(str(rng.choice(list("ACGT"))))
        else:
            out.append(c)
    return "".join(out) or "A"


def chain_case(rng, host):
    """A pair of related sequences and a list of chains (as the spec string of *_guide_from_chains)."""
    L = int(rng.integers(1500, 4000))
    q = rseq(rng, L, 0.002)
    r = mut(rng, q, rng.random() * 0.12)
    if rng.random() < 0.4:  # a long insertion: the "far" gap branch
        k = int(rng.integers(300, len(r) - 300))
        r = r[:k] + rseq(rng, int(rng.integers(800, 2500))) + r[k:]
    r = rseq(rng, int(rng.integers(0, 800))) + r + rseq(rng, int(rng.integers(0, 800)))
    q = rseq(rng, int(rng.integers(0, 800))) + q + rseq(rng, int(rng.integers(0, 800)))
    chains = [[a[:3] for a in c] for c in host.chains(q, r, 11)]
    chains = [c for c in chains if c]
    if not chains:
        return None
    chains.sort(key=lambda c: (c[0][0], c[0][1]))
    sel = [chains[0]]
    for c in chains[1:]:
        pq, pr = sel[-1][-1][0] + sel[-1][-1][2], sel[-1][-1][1] + sel[-1][-1][2]
        if c[-1][0] + c[-1][2] >= pq and c[-1][1] + c[-1][2] >= pr and c[0][0] > sel[-1][0][0] and \
                c[0][1] > sel[-1][0][1]:
            sel.append(c)
    spec = "|".join(";".join("%d,%d,%d" % a for a in c) for c in sel)
    return q, r, spec, int(rng.choice([0, 500]))


def make_genome(path, seed=1, glen=60000, nsd=6):
    """FASTA + .fai + bucket BED with planted duplications (fwd and rc), soft-masking and an N run."""
    rng = np.random.default_rng(seed)
    alpha = np.array(list("ACGT"))
    seq = alpha[rng.integers(0, 4, glen)].astype("U1")
    mask = np.zeros(glen, bool)
    i = 0
    while i < glen:
        L = int(rng.integers(50, 400))
        if rng.random() < 0.4:
            mask[i:i + L] = True
        i += L
    beds = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(nsd):
        L = int(rng.integers(1500, 6000))
        a = int(rng.integers(0, glen // 2 - L - 100))
        b = int(rng.integers(glen // 2, glen - L - 500))
        out = []
        div = rng.uniform(0.02, 0.12)
        for c in seq[a:a + L]:
            x = rng.random()
            if x < div * 0.6:
                out.append(alpha[rng.integers(0, 4)])
            elif x < div * 0.8:
                continue
            elif x < div:
                out.append(c)
                out.append(alpha[rng.integers(0, 4)])
            else:
                out.append(c)
        if rng.random() < 0.5:
            kk = int(rng.integers(100, len(out) - 100))
            out = out[:kk] + list(alpha[rng.integers(0, 4, int(rng.integers(50, 400)))]) + out[kk:]
        out = np.array(out)
        rcf = bool(rng.random() < 0.4)
        if rcf:
            out = np.array([comp[c] for c in out[::-1]])
        out = out[:glen - b]
        seq[b:b + len(out)] = out
        beds.append((a, a + L, b, b + len(out), rcf))
    seq[glen // 3:glen // 3 + 150] = "N"
    s = "".join(np.where(mask, np.char.lower(seq), seq))
    with open(path, "w") as f:
        f.write(">chrT test\n")
        for i in range(0, glen, 60):
            f.write(s[i:i + 60] + "\n")
    with open(path + ".fai", "w") as f:
        f.write("chrT\t%d\t%d\t60\t61\n" % (glen, len(">chrT test\n")))
    with open(path + ".bed", "w") as f:
        for (a, e, b, e2, rcf) in beds:
            f.write("chrT\t%d\t%d\tchrT\t%d\t%d\t\t\t+\t%s\t%d\t0\t\tOK\n" % (
                max(0, a - 300), min(glen, e + 300), max(0, b - 300), min(glen + 50, e2 + 300),
                "-" if rcf else "+", max(e - a, e2 - b)))
    return s, beds


def merge_case(rng):
    """Random seed hits as (BED lines for our library, spec lines for the reference driver)."""
    n = int(rng.integers(1, 40))
    lines, spec = [], []
    for _ in range(n):
        qn = str(rng.choice(["chr1", "chr2"]))
        rn = str(rng.choice(["chr1", "chr2", "chr3"]))
        qs, L = int(rng.integers(0, 5000)), int(rng.integers(100, 1500))
        rs, L2 = int(rng.integers(0, 5000)), int(rng.integers(100, 1500))
        rcf = int(rng.random() < 0.3)
        if rng.random() < 0.2 and lines:  # exact duplicates: ties in the sort key
            lines.append(lines[-1])
            spec.append(spec[-1])
            continue
        lines.append("%s\t%d\t%d\t%s\t%d\t%d\t\t\t+\t%s\t0\t0\t\t" % (qn, qs, qs + L, rn, rs, rs + L2,
                                                                       "-" if rcf else "+"))
        spec.append("%s %d %d %s %d %d %d" % (qn, qs, qs + L, rn, rs, rs + L2, rcf))
    return lines, "\n".join(spec)
