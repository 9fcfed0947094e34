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


def fasta_text(rng, nchr, line_blen):
    """A small multi-chromosome FASTA and its .fai entries (name, length, offset, line_blen, line_len)."""
    text, entries = "", []
    for c in range(nchr):
        length = int(rng.integers(1, 6 * line_blen + 40))
        if c == 0:
            length = 3 * line_blen  # ends exactly on a line boundary
        head = ">chr%d some description\n" % (c + 1)
        text += head
        entries.append(["chr%d" % (c + 1), length, len(text), line_blen, line_blen + 1])
        s = rseq(rng, length, 0.02)
        for i in range(0, length, line_blen):
            text += s[i:i + line_blen] + "\n"
    return text, entries


def fasta_queries(rng, length, line_blen):
    """(start, end) pairs incl. negative starts, ends beyond the chromosome (clamped), end=None, line-boundary starts."""
    qs = [(0, None), (0, length), (-3, length + 10), (0, 1), (length - 1, length), (length - 1, None)]
    for k in range(1, 4):
        if k * line_blen < length:
            qs += [(k * line_blen, None), (k * line_blen, min(length, k * line_blen + 1)), (k * line_blen - 1, length + 5)]
    for _ in range(12):
        a = int(rng.integers(-2, length))
        b = int(rng.integers(max(a, 0) + 1, length + 8))
        qs.append((a, b))
    return [(a, b) for (a, b) in qs if max(a, 0) < length]


# ---------------------------------------------------------------- stage-scale genome (BASELINE configs[0] / configs[2] shape)
_CODE2CHR = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate_codes(rng, s, d):
    """Vectorised form of mut(): 60 % substitution draws, 20 % deletions, 20 % insertions of the events."""
    r = rng.random(len(s))
    sub = r < d * 0.6
    dele = (r >= d * 0.6) & (r < d * 0.8)
    ins = (r >= d * 0.8) & (r < d)
    s = s.copy()
    s[sub] = rng.integers(0, 4, int(sub.sum()), dtype=np.uint8)
    cnt = np.ones(len(s), np.int64)
    cnt[dele] = 0
    cnt[ins] = 2
    out = np.repeat(s, cnt)
    pos = np.cumsum(cnt)[ins] - 1
    out[pos] = rng.integers(0, 4, len(pos), dtype=np.uint8)
    return out


def make_big_genome(path, seed=7, chrom_lens=(25_000_000, 18_000_000, 9_000_000, 500_000), n_pairs=5200,
                    slot=5000, min_len=1000, max_len=4000):
    """Multi-chromosome FASTA + .fai + seed BED (as `sedef search` would hand to `align bucket`): `n_pairs` planted
    duplications (forward and reverse-complement, 1-15 % divergence, some with a large insertion), 40 % soft-masking,
    N runs, and duplications planted against chromosome ends so that the extended hit is clamped by
    FastaReference::get_sequence (reference: src/fasta.cc:112-116, src/align_main.cc:303-321).
    Returns {name: genome string} and the number of seed lines."""
    rng = np.random.default_rng(seed)
    names = ["chr%s" % c for c in "ABCDEFGH"[:len(chrom_lens)]]
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in chrom_lens]
    # destination slots: distinct, `slot` bases each; the last slot of every chromosome is used first (end clamp)
    slots = [(c, k) for c, L in enumerate(chrom_lens) for k in range(L // slot)]
    last = [(c, L // slot - 1) for c, L in enumerate(chrom_lens)]
    first = [(c, 0) for c, L in enumerate(chrom_lens)]
    forced = last + first + [(c, L // slot - 2) for c, L in enumerate(chrom_lens)]
    rest = [s for s in slots if s not in set(forced)]
    pick = rng.permutation(len(rest))[:n_pairs - len(forced)]
    dests = forced + [rest[i] for i in pick]
    taken = set(dests)
    free = [s for s in slots if s not in taken]  # sources come from slots no copy is written to
    beds = []
    for k, (dc, ds) in enumerate(dests):
        L = int(rng.integers(min_len, max_len))
        sc, ss = free[int(rng.integers(0, len(free)))]
        sp = ss * slot + int(rng.integers(0, slot - L + 1))
        out = _mutate_codes(rng, seqs[sc][sp:sp + L], float(rng.uniform(0.01, 0.15)))
        if rng.random() < 0.3 and len(out) > 300:
            kk = int(rng.integers(100, len(out) - 100))
            out = np.concatenate([out[:kk], rng.integers(0, 4, int(rng.integers(50, 400)), dtype=np.uint8), out[kk:]])
        rcf = bool(rng.random() < 0.45)
        if k < len(forced):
            rcf = bool(k % 2)
        if rcf:
            out = (3 - out)[::-1]
        out = out[:slot]
        if k < len(last):  # flush against the chromosome end
            dp = chrom_lens[dc] - len(out)
        elif k < len(last) + len(first):  # flush against the chromosome start
            dp = 0
        else:
            dp = ds * slot + int(rng.integers(0, slot - len(out) + 1))
        seqs[dc][dp:dp + len(out)] = out
        j = int(rng.integers(0, 60))  # seeds are never exact
        beds.append((names[sc], sp + j, sp + L - j, names[dc], dp + j // 2, dp + len(out) - j, rcf))
    genome = {}
    with open(path, "wb") as f, open(path + ".fai", "w") as fai:
        off = 0
        for name, s, L in zip(names, seqs, chrom_lens):
            ch = _CODE2CHR[s]
            # soft-mask ~40 % in runs of 50..400, a few N runs
            runs = rng.integers(50, 400, L // 100 + 10)
            ends = np.cumsum(runs)
            ends = ends[ends < L]
            masked = rng.random(len(ends) + 1) < 0.4
            mask = np.repeat(masked, np.diff(np.concatenate(([0], ends, [L]))))
            ch = np.where(mask, ch + 32, ch).astype(np.uint8)
            for _ in range(max(2, L // 2_000_000)):
                p, n = int(rng.integers(0, L - 6000)), int(rng.integers(100, 5000))
                ch[p:p + n] = ord("N")
            head = (">%s synthetic\n" % name).encode()
            f.write(head)
            off += len(head)
            fai.write("%s\t%d\t%d\t60\t61\n" % (name, L, off))
            full = L // 60
            body = np.empty((full, 61), np.uint8)
            body[:, :60] = ch[:full * 60].reshape(full, 60)
            body[:, 60] = 10
            f.write(body.tobytes())
            off += full * 61
            if L % 60:
                f.write(ch[full * 60:].tobytes() + b"\n")
                off += L % 60 + 1
            genome[name] = ch.tobytes().decode()
    with open(path + ".seeds.bed", "w") as f:
        for (qn, qs, qe, rn, rs, re_, rcf) in beds:
            f.write("%s\t%d\t%d\t%s\t%d\t%d\t\t\t+\t%s\t%d\t0\t\tOK\n" % (qn, qs, qe, rn, rs, re_, "-" if rcf else "+",
                                                                        max(qe - qs, re_ - rs)))
    return genome, len(beds)


def _write_fasta(path, names, seqs, rng, n_runs_per_mb=0.5):
    """FASTA (60 columns) + .fai from code arrays: ~40 % soft-masked in runs of 50..400, a few N runs.  Returns
    {name: genome string}."""
    genome = {}
    with open(path, "wb") as f, open(path + ".fai", "w") as fai:
        off = 0
        for name, s in zip(names, seqs):
            L = len(s)
            ch = _CODE2CHR[s]
            runs = rng.integers(50, 400, L // 100 + 10)
            ends = np.cumsum(runs)
            ends = ends[ends < L]
            masked = rng.random(len(ends) + 1) < 0.4
            mask = np.repeat(masked, np.diff(np.concatenate(([0], ends, [L]))))
            ch = np.where(mask, ch + 32, ch).astype(np.uint8)
            for _ in range(max(2, int(L / 1e6 * n_runs_per_mb))):
                p, n = int(rng.integers(0, L - 6000)), int(rng.integers(100, 5000))
                ch[p:p + n] = ord("N")
            head = (">%s synthetic\n" % name).encode()
            f.write(head)
            off += len(head)
            fai.write("%s\t%d\t%d\t60\t61\n" % (name, L, off))
            full = L // 60
            body = np.empty((full, 61), np.uint8)
            body[:, :60] = ch[:full * 60].reshape(full, 60)
            body[:, 60] = 10
            f.write(body.tobytes())
            off += full * 61
            if L % 60:
                f.write(ch[full * 60:].tobytes() + b"\n")
                off += L % 60 + 1
            genome[name] = ch.tobytes().decode()
    return genome


def make_chr1_genome(path, seed=13, chrom_lens=(249_000_000, 6_000_000), n_pairs=2100, slot=125_000, small=12_500):
    """BASELINE configs[2] at size: a chr1-sized chromosome (249 Mb) and a small second one with
    `n_pairs` planted duplications of 1-100 kb (log-uniform) at 2-25 % divergence -- forward and reverse-complement, a
    third with one or two large indels (50-3000 bases), every tenth a TANDEM copy on the same chromosome and strand right
    behind its source (the extended hits overlap: src/chain.cc:67-69, src/refine.cc:42-53), copies flush against both
    chromosome ends -- plus FASTA, .fai and the seed BED `sedef search` would hand to `align bucket`.  Long duplications
    mean many chains per pair, far gaps between hits of a refined path (src/refine.cc:77), hits extended to the 15 kb cap
    (src/hit.cc:200-207) and DP tasks of 10^7 cells inside the stage.  Returns ({name: genome string}, seed lines)."""
    rng = np.random.default_rng(seed)
    names = ["chr1", "chr1b"][:len(chrom_lens)]
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in chrom_lens]
    # regions: slots of `slot` bases for copies of more than 10 kb, a ninth of them cut into slots of `small` bases for the
    # short ones; (chromosome, start, size), in random order
    big = [(c, k * slot, slot) for c, L in enumerate(chrom_lens) for k in range(1, L // slot - 1)]
    ends = [(c, (L // slot - 1) * slot, L - (L // slot - 1) * slot) for c, L in enumerate(chrom_lens)] + \
           [(c, 0, slot) for c, L in enumerate(chrom_lens)]
    big = [big[i] for i in rng.permutation(len(big))]
    n_cut = len(big) // 9
    tiny = [(c, st + q * small, small) for (c, st, _) in big[:n_cut] for q in range(slot // small)]
    tiny = [tiny[i] for i in rng.permutation(len(tiny))]
    big = big[n_cut:]
    beds = []
    k = 0
    while len(beds) < n_pairs:
        L = int(np.exp(rng.uniform(np.log(1000), np.log(100_000))))
        d = float(rng.uniform(0.02, 0.25))
        if k < len(ends):  # (the copies at the chromosome ends must survive the chaining: moderate length and divergence)
            L, d = int(rng.integers(12_000, 40_000)), float(rng.uniform(0.02, 0.08))
        pool = tiny if L <= small - 2500 else big
        if len(pool) < 2:
            break
        sc, s0, ssz = pool.pop()
        tandem = k % 10 == 9 and 2 * L + 4500 < ssz
        sp = s0 if tandem else s0 + int(rng.integers(0, ssz - L + 1))
        out = _mutate_codes(rng, seqs[sc][sp:sp + L], d)
        if rng.random() < 0.33 and len(out) > 600:
            for _ in range(int(rng.integers(1, 3))):
                if len(out) <= 600:
                    break
                kk = int(rng.integers(200, len(out) - 200))
                n_ind = int(rng.integers(50, 3000))
                if rng.random() < 0.5:
                    out = np.concatenate([out[:kk], rng.integers(0, 4, n_ind, dtype=np.uint8), out[kk:]])
                else:
                    out = np.concatenate([out[:kk], out[min(len(out) - 100, kk + n_ind):]])
        rcf = bool(rng.random() < 0.45) and not tandem
        if k < len(ends):
            rcf = bool(k % 2)
        if tandem:  # same chromosome, same strand, right behind the source
            out = out[:ssz - L - 1500]
            dc, dp = sc, sp + L + int(rng.integers(0, 1500))
        elif k < len(ends):  # flush against a chromosome end / start
            dc, d0, dsz = ends[k]
            out = out[:dsz]
            dp = chrom_lens[dc] - len(out) if d0 else 0
        else:
            dc, d0, dsz = pool.pop()
            out = out[:dsz]
            dp = d0 + int(rng.integers(0, dsz - len(out) + 1))
        if rcf:
            out = (3 - out)[::-1]
        seqs[dc][dp:dp + len(out)] = out
        j = int(rng.integers(0, 60))  # seeds are never exact
        beds.append((names[sc], sp + j, sp + L - j, names[dc], dp + j // 2, dp + len(out) - j, rcf))
        k += 1
    genome = _write_fasta(path, names, seqs, rng, n_runs_per_mb=0.2)
    with open(path + ".seeds.bed", "w") as f:
        for (qn, qs, qe, rn, rs, re_, rcf) in beds:
            f.write("%s\t%d\t%d\t%s\t%d\t%d\t\t\t+\t%s\t%d\t0\t\tOK\n" % (qn, qs, qe, rn, rs, re_, "-" if rcf else "+",
                                                                        max(qe - qs, re_ - rs)))
    return genome, len(beds)


def chunk_case(seed=5, n=60050, d=0.03):
    """Two related sequences longer than Align::MAX_KSW_SEQ_LEN = 60,000 (reference: src/globals.h:54): align_helper
    cuts them into 60 kb x 60 kb chunks at equal offsets (src/align.cc:46-57)."""
    rng = np.random.default_rng(seed)
    a = _mutate_codes(rng, rng.integers(0, 4, n, dtype=np.uint8), 0.0)
    b = _mutate_codes(rng, a, d)
    low = rng.random(len(a)) < 0.3
    sa = np.where(low, _CODE2CHR[a] + 32, _CODE2CHR[a]).astype(np.uint8).tobytes().decode()
    return sa, _CODE2CHR[b].tobytes().decode()


def far_equal_case(rng, gap):
    """Two chains separated by unrelated stretches of EQUAL length > 1000 in both sequences: the far-gap branch of the
    reference (src/align.cc:131-139) then pushes a run of length zero, which stays in the CIGAR's run list."""
    a = rseq(rng, int(rng.integers(300, 600)))
    b = rseq(rng, int(rng.integers(300, 600)))
    q = a + rseq(rng, gap) + b
    r = mut(rng, a, 0.0) + rseq(rng, gap) + mut(rng, b, 0.0)
    la, lb = len(a), len(b)
    spec = "0,0,%d|%d,%d,%d" % (la, la + gap, la + gap, lb)
    return q, r, spec, int(rng.choice([0, 500]))


# ---------------------------------------------------------------- candidate pairs for the stage-level fixtures
def _render(rng, codes, mask_frac=0.4, n_runs=0, lower_all=False):
    """Code array -> FASTA characters: soft-masked runs of 50..400 (~mask_frac of the bases), `n_runs` short N runs."""
    L = len(codes)
    ch = _CODE2CHR[codes].copy()
    if lower_all:
        ch = ch + 32
    elif mask_frac > 0 and L:
        runs = rng.integers(50, 400, L // 50 + 4)
        ends = np.cumsum(runs)
        ends = ends[ends < L]
        masked = rng.random(len(ends) + 1) < mask_frac
        mask = np.repeat(masked, np.diff(np.concatenate(([0], ends, [L]))))
        ch = np.where(mask, ch + 32, ch)
    ch = ch.astype(np.uint8)
    for _ in range(n_runs):
        if L > 40:
            p, n = int(rng.integers(0, L - 20)), int(rng.integers(1, 20))
            ch[p:p + n] = ord("N")
    return ch.tobytes().decode()


def _rand(rng, n):
    return rng.integers(0, 4, int(n), dtype=np.uint8)


def _indels(rng, out, count, lo=50, hi=3000):
    for _ in range(count):
        if len(out) <= 600:
            break
        kk = int(rng.integers(200, len(out) - 200))
        n_ind = int(rng.integers(lo, hi))
        if rng.random() < 0.5:
            out = np.concatenate([out[:kk], _rand(rng, n_ind), out[kk:]])
        else:
            out = np.concatenate([out[:kk], out[min(len(out) - 100, kk + n_ind):]])
    return out


STAGE_PAIR_KINDS = ("plain", "indel", "rc", "same_chr", "self_overlap", "tandem", "far_equal", "far_unequal", "far_cut",
                    "threshold", "dp_tie", "low_upper", "n_runs", "long")


def stage_pair_case(rng, kind):
    """One candidate pair as fast_align sees it (src/chain.cc:203: two strings + the seed hit's names, strands, starts).
    Returns dict(query, ref, qname, rname, q_rc, r_rc, qstart, rstart, kind)."""
    d = float(rng.uniform(0.01, 0.18))
    qname, rname, r_rc = "chr1", "chr2", False
    qstart, rstart = int(rng.integers(0, 5_000_000)), int(rng.integers(0, 5_000_000))
    fl = lambda lo=0, hi=1500: _rand(rng, rng.integers(lo, hi))
    mask, nr = 0.4, 0
    if kind in ("plain", "indel", "rc", "same_chr", "low_upper", "n_runs", "long"):
        L = int(np.exp(rng.uniform(np.log(1000), np.log(9000))))
        if kind == "long":
            L = int(rng.integers(14000, 26000))
        u = _rand(rng, L)
        c = _mutate_codes(rng, u, d)
        if kind in ("indel", "long") or rng.random() < 0.25:
            c = _indels(rng, c, int(rng.integers(1, 4 if kind != "long" else 7)))
        q = np.concatenate([fl(), u, fl()])
        r = np.concatenate([fl(), c, fl()])
        if kind == "rc":
            r_rc, rname = True, str(rng.choice(["chr1", "chr2"]))
        if kind == "same_chr":
            rname = "chr1"
            rstart = qstart + len(q) + int(rng.integers(0, 3000))
        if kind == "low_upper":
            mask = 0.97
        if kind == "n_runs":
            nr = int(rng.integers(2, 12))
        qs, rs = _render(rng, q, mask, nr), _render(rng, r, mask, nr)
    elif kind == "self_overlap":
        # one region of a chromosome holding a tandem array; query and ref are overlapping windows of it on one strand
        unit = _rand(rng, rng.integers(700, 2200))
        parts = [fl(200, 900)]
        for _ in range(int(rng.integers(3, 6))):
            parts += [_mutate_codes(rng, unit, float(rng.uniform(0.01, 0.1))), fl(0, 300)]
        parts.append(fl(200, 900))
        g = _render(rng, np.concatenate(parts), mask)
        G = len(g)
        a0, a1 = 0, int(rng.integers(G // 2, G))
        b0 = int(rng.integers(1, max(2, a1 - 600)))
        b1 = G
        qs, rs = g[a0:a1], g[b0:b1]
        rname = "chr1"
        rstart = qstart + b0
    elif kind == "tandem":
        unit = _rand(rng, rng.integers(900, 3000))
        cp = lambda: _mutate_codes(rng, unit, float(rng.uniform(0.01, 0.12)))
        q = np.concatenate([fl()] + [x for _ in range(int(rng.integers(1, 4))) for x in (cp(), fl(0, 250))] + [fl()])
        r = np.concatenate([fl()] + [x for _ in range(int(rng.integers(2, 4))) for x in (cp(), fl(0, 250))] + [fl()])
        qs, rs = _render(rng, q, mask), _render(rng, r, mask)
    elif kind in ("far_equal", "far_unequal", "far_cut"):
        a, b = _rand(rng, rng.integers(600, 2500)), _rand(rng, rng.integers(600, 2500))
        dd = float(rng.uniform(0.0, 0.08))
        if kind == "far_equal":
            g1 = g2 = int(rng.integers(1001, 3000))
        elif kind == "far_unequal":
            g1, g2 = int(rng.integers(0, 4000)), int(rng.integers(1001, 9000))
            if rng.random() < 0.5:
                g1, g2 = g2, g1
        else:
            g1, g2 = int(rng.integers(0, 3000)), int(rng.integers(9900, 10200))
            if rng.random() < 0.5:
                g1, g2 = g2, g1
        q = np.concatenate([fl(0, 700), a, _rand(rng, g1), b, fl(0, 700)])
        r = np.concatenate([fl(0, 700), _mutate_codes(rng, a, dd), _rand(rng, g2), _mutate_codes(rng, b, dd), fl(0, 700)])
        qs, rs = _render(rng, q, mask), _render(rng, r, mask)
    elif kind == "threshold":
        # an island of exact match whose chain span lands on either side of Search::MIN_READ_SIZE * (1 - MAX_ERROR) =
        # 489.99999999999994 (src/chain.cc:233-238), inside a copy too diverged for seeds (substitutions only, so that the
        # side extension of a kept chain still runs through it); the bases next to the island differ
        isl = int(rng.choice([486, 488, 489, 489, 489, 490, 490, 490, 491, 493]))
        core = _rand(rng, isl)
        lf, rf = _rand(rng, rng.integers(350, 800)), _rand(rng, rng.integers(350, 800))

        def subst(x, dv):
            y = x.copy()
            hit = rng.random(len(x)) < dv
            y[hit] = (y[hit] + rng.integers(1, 4, int(hit.sum()), dtype=np.uint8)) % 4
            return y
        dv = float(rng.uniform(0.34, 0.4))
        lf2, rf2 = subst(lf, dv), subst(rf, dv)
        lf2[-1] = (lf[-1] + 1) % 4
        rf2[0] = (rf[0] + 1) % 4
        q = np.concatenate([fl(0, 400), lf, core, rf, fl(0, 400)])
        r = np.concatenate([fl(0, 400), lf2, core, rf2, fl(0, 400)])
        lower = bool(rng.random() < 0.7)
        qs, rs = _render(rng, q, 0.0, 0, lower), _render(rng, r, 0.0, 0, lower)
    elif kind == "dp_tie":
        # an exact island P of L bases, `g` unrelated bases in BOTH sequences, then a long copy C: joining P to C costs
        # int(1 * g) + int(100 + 0.5 * 0) and P scores 10 * L, so 10 L == g + 100 makes `sco >= dp[ai]` an equality
        # (src/refine.cc:91-96)
        L = int(rng.integers(95, 140))
        g = 10 * L - 100 + int(rng.choice([0, 0, 0, 1, -1]))
        p = _rand(rng, L)
        c = _rand(rng, rng.integers(1200, 3000))
        x, y = _rand(rng, g), _rand(rng, g)
        # P's match must not grow by chance: the bases around it differ
        lq, lr = _rand(rng, 300), _rand(rng, 300)
        lr[-1] = (lq[-1] + 1) % 4
        y[0] = (x[0] + 1) % 4
        y[-1] = (x[-1] + 1) % 4
        cc = c.copy()
        q = np.concatenate([lq, p, x, c, fl(0, 300)])
        r = np.concatenate([lr, p, y, cc, fl(0, 300)])
        qs, rs = _render(rng, q, 0.0), _render(rng, r, 0.0)
    else:
        raise ValueError(kind)
    return dict(kind=kind, query=qs, ref=rs, qname=qname, rname=rname, q_rc=False, r_rc=bool(r_rc), qstart=qstart,
                rstart=rstart)


def make_stage_fixture(seed, chrom_lens=(90_000, 50_000, 24_000), line_blens=(60, 50, 70), n_pairs=22, max_len=9000):
    """A small genome as TEXT (FASTA, .fai) and a bucket file as `sedef align bucket` writes them (extended seeds:
    Hit::extend does not clamp the ends, src/hit.cc:200-207), for the stage-level fixtures: planted duplications forward
    and reverse-complement, tandem copies on one chromosome and strand (overlapping extended hits), copies flush against
    chromosome starts and ends (FastaReference::get_sequence clamps, src/fasta.cc:112-116, before the rc remap uses the
    clamped end, src/align_main.cc:317-321), seed lines of 10 to 15 fields, complexity classes (src/align_main.cc:243-264)
    in shuffled file order."""
    rng = np.random.default_rng(seed)
    names = ["chrA", "chrB", "chrC"][:len(chrom_lens)]
    seqs = [_rand(rng, L) for L in chrom_lens]
    cursor = [300] * len(chrom_lens)  # next free position per chromosome for sources / copies
    beds = []

    def take(c, n):
        p = cursor[c]
        if p + n + 200 > chrom_lens[c]:
            return None
        cursor[c] = p + n + int(rng.integers(200, 1200))
        return p
    k = 0
    while len(beds) < n_pairs and k < 10 * n_pairs:
        k += 1
        L = int(np.exp(rng.uniform(np.log(900), np.log(max_len))))
        d = float(rng.uniform(0.01, 0.16))
        sc = int(rng.integers(0, len(names)))
        dc = int(rng.integers(0, len(names)))
        mode = ["plain", "rc", "tandem", "end", "start"][len(beds) % 5] if len(beds) < 10 else str(rng.choice(["plain", "rc", "tandem"]))
        sp = take(sc, L)
        if sp is None:
            continue
        out = _mutate_codes(rng, seqs[sc][sp:sp + L], d)
        if rng.random() < 0.35:
            out = _indels(rng, out, int(rng.integers(1, 3)), 50, 1500)
        rcf = mode == "rc" or (mode in ("end", "start") and rng.random() < 0.5)
        if mode == "tandem":
            dc = sc
            dp = take(dc, len(out))
        elif mode == "end":
            dp = chrom_lens[dc] - len(out)
            if cursor[dc] + 200 > dp:
                continue
        elif mode == "start":
            if any(b[3] == names[dc] and b[4] < 300 + len(out) for b in beds) or len(out) > 290:
                out = out[:290]
            dp = 0
        else:
            dp = take(dc, len(out))
        if dp is None:
            continue
        if rcf:
            out = (3 - out)[::-1]
        seqs[dc][dp:dp + len(out)] = out
        j = int(rng.integers(0, 40))
        qs, qe, rs, re_ = sp + j, sp + L - j, dp + j // 2, dp + len(out) - j // 3
        w = min(2500, int(1.0 * max(qe - qs, re_ - rs)))  # Hit::extend(factor, max_extend) with small values: short windows
        beds.append([names[sc], max(0, qs - w), qe + w, names[dc], max(0, rs - w), re_ + w, rcf, mode])
    fasta, fai, off = "", "", 0
    for name, s, lb in zip(names, seqs, line_blens):
        text = _render(rng, s, 0.4, 2)
        head = ">%s fixture chromosome\n" % name
        fasta += head
        off += len(head)
        fai += "%s\t%d\t%d\t%d\t%d\n" % (name, len(text), off, lb, lb + 1)
        body = "".join(text[i:i + lb] + "\n" for i in range(0, len(text), lb))
        fasta += body
        off += len(body)
    order = rng.permutation(len(beds))
    lines = []
    for n_, i in enumerate(order):
        qn, qs, qe, rn, rs, re_, rcf, mode = beds[i]
        f = [qn, str(qs), str(qe), rn, str(rs), str(re_), "", "", "+", "-" if rcf else "+"]
        nf = [13, 10, 14, 15, 13][n_ % 5]
        if nf >= 13:
            f += [str(max(qe - qs, re_ - rs)), "0", ""]
        if nf >= 14:
            f += [str(int(rng.integers(0, 50)))]
        if nf >= 15:
            f += ["seed%d" % n_]
        if nf == 13 and n_ % 2:
            f[6] = "name%d" % n_
        lines.append("\t".join(f))
    return dict(fasta=fasta, fai=fai, bed="\n".join(lines) + "\n", modes=[b[7] for b in beds])
