"""f4, host side: `sedef stats generate` (reference: src/stats_main.cc:33-336), the consumer of `align generate`'s BEDPE.

src/stats_main.cc cannot be compiled here (boost/dynamic_bitset.hpp): the text side is a restatement, parity unpinned.
What is checked:
  * the number formatting against the reference's own vendored fmt (fmt 4.0.1 "{}" of a double), live + fixed values;
  * the whole table against an independent statement of the reference's algorithm on COLUMN STRINGS, written in this
    file from src/stats_main.cc (subhit / gap_split / split_alignment / process) and src/align.cc (trims, cigar from
    columns) -- the product works on run-length CIGARs and takes the counters from the device kernel;
  * on the GPU: the device's counters give the same bytes as the oracle's column walk, through the library and the CLI."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import hostgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from sedef_amd import host as h
    from sedef_amd.build import build_library
    build_library()
    h.build_host()
    return h


def _cols_hook(oracle):
    oracle.lib.sdfo_stats_columns.restype = C.c_int
    return C.cast(oracle.lib.sdfo_stats_columns, C.c_void_p)


FIXED = [(0.0, "0"), (1.0, "1"), (0.5, "0.5"), (1.0 / 3, "0.333333"), (0.934210526, "0.934211"), (123456789.0, "1.23457e+08"),
         (1e-5, "1e-05"), (0.000123456789, "0.000123457"), (float("inf"), "inf"), (float("-inf"), "-inf"), (0.75, "0.75"),
         (100.0, "100"), (0.1 + 0.2, "0.3"), (1234567.0, "1.23457e+06"), (0.999999949, "1")]


def test_double_formatting_fixed_values(host):
    for x, s in FIXED:  # (what the reference's fmt printed for them: tests/test_stats_generate.py history, oracle/_ref)
        assert host.format_double(x) == s
    assert host.format_double(float("nan")).lstrip("-") == "nan"


def test_double_formatting_equals_reference_fmt_live(host):
    from oracle.binding import ReferenceAlign
    try:
        ref = ReferenceAlign()
    except (FileNotFoundError, OSError):
        pytest.skip("reference checkout / oracle/_ref/libref_align.so not available")
    rng = np.random.default_rng(3)
    xs = [x for x, _ in FIXED] + [-0.75 * math.log(1.0 - 4.0 / 3 * p) for p in rng.random(200) * 0.74] + \
        list(rng.random(300)) + list(10.0 ** rng.uniform(-12, 12, 300)) + [1 - k / 997.0 for k in range(0, 997, 7)]
    for x in xs:
        assert host.format_double(x) == ref.fmt_double(x), x
    assert host.format_double(math.log(-1.0) if False else float("nan")).lstrip("-") == ref.fmt_double(float("nan")).lstrip("-")


# ---- the reference's algorithm on column strings (independent of the product's run-length form) ----------------------
MATCH, MISMATCH, GAP_OPEN, GAP_EXTEND = 5, -4, -40, -1


def _ceq(a, b):  # src/align.cc:29-35
    return a.upper() == b.upper() and a.upper() != "N" and a != "-"


class ColAln:
    """Alignment as the reference holds it: the bases it covers and its CIGAR; columns on demand (src/align.cc:274-315)."""

    def __init__(self, a, b, cigar):
        self.a, self.b, self.cigar = a, b, list(cigar)

    def columns(self):
        ca, cb, ia, ib = [], [], 0, 0
        for op, n in self.cigar:
            for _ in range(n):
                if op != "D":
                    cb.append(self.b[ib]); ib += 1
                else:
                    cb.append("-")
                if op != "I":
                    ca.append(self.a[ia]); ia += 1
                else:
                    ca.append("-")
        return "".join(ca), "".join(cb)

    def counters(self):
        ca, cb = self.columns()
        gaps = sum(1 for op, n in self.cigar if op != "M")
        gap_bases = sum(n for op, n in self.cigar if op != "M")
        m = sum(1 for x, y in zip(ca, cb) if x != "-" and y != "-" and _ceq(x, y))
        mm = sum(1 for x, y in zip(ca, cb) if x != "-" and y != "-" and not _ceq(x, y))
        return gaps, gap_bases, mm, m

    def _score_steps(self, ca, cb, order):
        n = len(ca)
        for i in order:
            if ca[i] != "-" and cb[i] != "-" and _ceq(ca[i], cb[i]):
                yield i, MATCH
            elif ca[i] != "-" and cb[i] != "-":
                yield i, MISMATCH
            else:
                j = i + 1 if order.step < 0 else i - 1
                first = i == (n - 1 if order.step < 0 else 0)
                opens = first or (ca[i] == "-" and ca[j] != "-") or (cb[i] == "-" and cb[j] != "-")
                yield i, (GAP_OPEN if opens else 0) + GAP_EXTEND

    def trim_back(self):  # src/align.cc:400-456
        ca, cb = self.columns()
        best, best_i, score = 0, -1, 0
        for i, s in self._score_steps(ca, cb, range(len(ca))):
            score += s
            if score >= best:
                best, best_i = score, i
        self._keep(ca, cb, 0, best_i + 1)

    def trim_front(self):  # src/align.cc:343-398 (its "nothing found" marker is len(a))
        ca, cb = self.columns()
        best, best_i, score = 0, len(self.a), 0
        for i, s in self._score_steps(ca, cb, range(len(ca) - 1, -1, -1)):
            score += s
            if score >= best:
                best, best_i = score, i
        if best_i == len(self.a):
            self.a, self.b, self.cigar = "", "", []
            return
        self._keep(ca, cb, best_i, len(ca))

    def _keep(self, ca, cb, lo, hi):
        ca, cb = ca[lo:hi], cb[lo:hi]
        self.a, self.b = ca.replace("-", ""), cb.replace("-", "")
        self.cigar = _cigar_from_columns(ca, cb) if ca else []


def _cigar_from_columns(ca, cb):  # src/align.cc:480-501
    out = []
    for x, y in zip(ca, cb):
        op = "I" if x == "-" else "D" if y == "-" else "M"
        if out and out[-1][0] == op:
            out[-1][1] += 1
        else:
            out.append([op, 1])
    return [(o, n) for o, n in out]


def _subhit(h, start, end):  # src/stats_main.cc:33-84
    ca, cb = h["aln"].columns()
    end = min(end, len(ca))
    if start >= end:
        return None
    sa = sum(1 for c in ca[:start] if c != "-")
    sb = sum(1 for c in cb[:start] if c != "-")
    la = sum(1 for c in ca[start:end] if c != "-")
    lb = sum(1 for c in cb[start:end] if c != "-")
    al = ColAln(ca[start:end].replace("-", ""), cb[start:end].replace("-", ""), _cigar_from_columns(ca[start:end], cb[start:end]))
    al.trim_back()
    al.trim_front()
    n = dict(h)
    n["aln"] = al
    n["qs"] = h["qs"] + sa
    n["qe"] = n["qs"] + la
    if h["rc"]:
        n["rs"], n["re"] = h["re"] - (lb + sb), h["re"] - sb
    else:
        n["rs"] = h["rs"] + sb
        n["re"] = n["rs"] + lb
    return n


def _gap_split(h, max_ok_gap, min_split):  # src/stats_main.cc:86-161
    al = h["aln"]
    gaps, sa, sb, col = [], 0, 0, 0
    for op, n in al.cigar:
        if n and op != "M":
            gaps.append((n, sa, sb, n if op == "D" else 0, n if op != "D" else 0, col))
        if op != "D":
            sb += n
        if op != "I":
            sa += n
        col += n
    if max_ok_gap > -1:
        g_, gb_, mm_, m_ = al.counters()
        # (equal lengths: the reference's std::sort order is libstdc++'s; the cases below have one longest gap)
        for n, ga, gb, la, lb, start in sorted(gaps, key=lambda g: -g[0]):
            if ga < min_split or gb < min_split:
                continue
            if len(al.a) - (ga + la) < min_split or len(al.b) - (gb + lb) < min_split:
                continue
            if 100.0 * n / (m_ + gb_ + mm_) >= max_ok_gap:
                out = []
                for part in (_subhit(h, 0, start), _subhit(h, start + n, 10 ** 9)):
                    if part:
                        out += _gap_split(part, max_ok_gap, min_split)
                return out
    return [h]


def _split_alignment(h, max_ok_gap, min_split):  # src/stats_main.cc:163-211
    ca, cb = h["aln"].columns()
    hits, pa, pb, begin = [], 0, 0, 0
    for i in range(len(ca)):
        for which in (0, 1):
            c = (ca, cb)[which][i]
            prev = pa if which == 0 else pb
            if c.upper() == "N":
                prev += 1
            else:
                if prev >= 100:
                    part = _subhit(h, begin, i - prev)
                    if part:
                        hits.append(part)
                    begin = i
                prev = 0
            if which == 0:
                pa = prev
            else:
                pb = prev
    if not begin:
        hits.append(h)
    else:
        part = _subhit(h, begin, len(ca))
        if part:
            hits.append(part)
    out = []
    for x in hits:
        out += _gap_split(x, max_ok_gap, min_split)
    return out


_RC = {ord(a): b for a, b in zip("ACGTNacgtn", "TGCANtgcan")}


def _rc(s):  # src/util.cc:43-48 with src/common.h:72-87: the case stays, anything else becomes N
    return "".join(_RC.get(ord(c), "N") for c in reversed(s))


def _g(x):
    return "%g" % x


def model_table(genome, bed_lines, max_ok_gap=-1, min_split=1000, uppercase=100, max_error=0.5):
    """`stats generate` on column strings; returns the data lines."""
    hits = []
    for ln in bed_lines:
        f = ln.rstrip("\n").split("\t")
        qn, qs, qe, rn, rs, re_ = f[0], int(f[1]), int(f[2]), f[3], int(f[4]), int(f[5])
        rcf, cigar = f[9][0] != "+", f[12]
        if (qn, qs, qe) > (rn, rs, re_):
            qn, rn, qs, rs, qe, re_ = rn, qn, rs, qs, re_, qe
            cigar = cigar.translate(str.maketrans("ID", "DI"))
        hits.append(dict(qn=qn, qs=qs, qe=qe, rn=rn, rs=rs, re=re_, rc=rcf, cigar=cigar))
    hits.sort(key=lambda h: (h["rc"], h["qn"], h["rn"], h["qs"], h["rs"]))
    import re
    out = []
    for h in hits:
        h["qe"], h["re"] = min(h["qe"], len(genome[h["qn"]])), min(h["re"], len(genome[h["rn"]]))
        fa, fb = genome[h["qn"]][h["qs"]:h["qe"]], genome[h["rn"]][h["rs"]:h["re"]]
        if h["rc"]:
            fb = _rc(fb)
        h["aln"] = ColAln(fa, fb, [(o, int(n)) for n, o in re.findall(r"(\d+)([MID])", h["cigar"])])
        for p in _split_alignment(h, max_ok_gap, min_split):
            ca, cb = p["aln"].columns()
            if len(ca) < 900:
                continue
            A, B = ca.upper(), cb.upper()
            indel_a, indel_b = A.count("-"), B.count("-")
            both = [(x, y, xo, yo) for x, y, xo, yo in zip(A, B, ca, cb) if x != "-" and y != "-"]
            alnB = len(both)
            matchB = sum(1 for x, y, _, _ in both if x == y)
            mismatchB = alnB - matchB
            pur = "AG"
            trans = sum(1 for x, y, _, _ in both if x != y and ((x in pur and y in pur) or (x not in pur and y in "CT")))
            transv = mismatchB - trans
            upA = sum(1 for c in ca if c != "-" and c.upper() != "N" and c.isupper())
            upB = sum(1 for c in cb if c != "-" and c.upper() != "N" and c.isupper())
            upM = sum(1 for x, y, xo, yo in both if x == y and xo.isupper() and yo.isupper())
            gaps, gap_bases, mm, m = p["aln"].counters()
            with np.errstate(all="ignore"):
                d = np.float64
                fracMatch, fracMatchIndel = d(matchB) / d(alnB), d(matchB) / d(len(ca))
                jcK = d(-0.75) * np.log(d(1.0) - d(4.0) / d(3) * (d(mismatchB) / d(alnB)))
                pp, qq = d(trans) / d(alnB), d(transv) / d(alnB)
                k2K = d(0.5) * np.log(d(1.0) / (1 - d(2.0) * pp - qq)) + d(0.25) * np.log(d(1.0) / (1 - d(2.0) * qq))
            same = p["qn"] == p["rn"] and not p["rc"]
            ov = max(0, min(p["qe"], p["re"]) - max(p["qs"], p["rs"])) if same else 0
            too_big = same and ((p["qe"] - p["qs"] - ov) < 100 or (p["re"] - p["rs"] - ov) < 100)
            scaled = (gaps + mm) / float(gaps + mm + m)
            if not (upA >= uppercase and upB >= uppercase and not too_big and scaled <= max_error and upM >= uppercase):
                continue
            tot = float(m + gap_bases + mm)
            cig = "".join("%d%s" % (n, o) for o, n in p["aln"].cigar if n)
            bed = [p["qn"], p["qs"], p["qe"], p["rn"], p["rs"], p["re"], "S", "%.1f" % (100.0 * mm / tot + 100.0 * gap_bases / tot),
                   "+", "-" if p["rc"] else "+", max(p["qe"] - p["qs"], p["re"] - p["rs"]), len(ca),
                   "m=%.1f;g=%.1f" % (100.0 * mm / tot, 100.0 * gap_bases / tot)]
            row = bed + [indel_a, indel_b, alnB, matchB, mismatchB, trans, transv, _g(fracMatch), _g(fracMatchIndel), _g(jcK), _g(k2K),
                         gaps, upA, upB, upM, m, mm, gaps, gap_bases, cig, _g(1 - scaled)]
            out.append("\t".join(str(x) for x in row))
    return out


def _stage(host, oracle, tmp_path, seed, n_gap_runs=0):
    """A small genome with planted duplications -> `align generate` (oracle DP) -> BEDPE; optionally runs of N planted
    INSIDE duplicated regions of both copies (assembly gaps the alignments span)."""
    fa = str(tmp_path / "genome.fa")
    genome, beds = hostgen.make_genome(fa, seed=seed, glen=220000, nsd=20)
    bed = str(tmp_path / "final.bed")
    hook = C.cast(oracle.lib.sdfo_extz2, C.c_void_p)
    host.generate(fa, fa + ".bed", 11, bed, test_dp=hook)
    return fa, genome, bed


def test_stats_table_equals_column_string_model(host, oracle, tmp_path):
    fa, genome, bed = _stage(host, oracle, tmp_path, seed=21)
    out = str(tmp_path / "stats.tsv")
    lines, hits, pieces, columns = host.stats_generate(fa, bed, out, test_cols=_cols_hook(oracle))
    text = open(out).read().splitlines()
    assert text[0].startswith("#chr1\tstart1\tend1\tchr2") and text[0].count("\t") == 33
    assert lines == len(text) - 1 >= 10 and hits >= lines and columns >= 900 * lines
    gen = genome if isinstance(genome, dict) else {"chrT": genome}
    assert text[1:] == model_table(gen, open(bed).read().splitlines())
    # the cut at large gaps (off by default): a threshold low enough to fire
    out2 = str(tmp_path / "stats_gap.tsv")
    host.stats_generate(fa, bed, out2, max_ok_gap=1, min_split=300, test_cols=_cols_hook(oracle))
    t2 = open(out2).read().splitlines()
    assert t2[1:] == model_table(gen, open(bed).read().splitlines(), max_ok_gap=1, min_split=300)
    assert t2[1:] != text[1:]


def _handmade(tmp_path, rng, n_cases=12):
    """A genome and BEDPE lines written by hand: copies with substitutions and a few indels whose alignments (CIGARs built
    here) span runs of 100+ N in one or both sequences -- assembly gaps, where `stats generate` cuts (src/stats_main.cc:
    163-199) --, on both strands, soft-masked."""
    alpha = np.frombuffer(b"ACGT", np.uint8)
    chunks, bed, pos = [], [], 0
    comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}

    def put(arr):
        nonlocal pos
        chunks.append(arr)
        pos += len(arr)
        return pos - len(arr)

    put(alpha[rng.integers(0, 4, 500)])
    for k in range(n_cases):
        L = int(rng.integers(2400, 5000))
        a = alpha[rng.integers(0, 4, L)].copy()
        ops, b_parts, i = [], [], 0
        while i < L:  # blocks of matches with substitutions, separated by short indels
            n = min(L - i, int(rng.integers(300, 900)))
            blk = a[i:i + n].copy()
            sub = rng.random(n) < 0.04
            blk[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
            b_parts.append(blk)
            ops.append(("M", n))
            i += n
            if i < L and rng.random() < 0.6:
                g = int(rng.integers(1, 30))
                if rng.random() < 0.5 and i + g < L:
                    ops.append(("D", g))  # bases of a only
                    i += g
                else:
                    ops.append(("I", g))
                    b_parts.append(alpha[rng.integers(0, 4, g)])
        b = np.concatenate(b_parts)
        # assembly gaps: N over the same columns of both copies (k % 3 == 0), of a only (1), two runs (2)
        cols_a = np.concatenate([np.arange(n) if o != "I" else np.full(n, -1) for o, n in ops])
        cols_a = np.where(cols_a >= 0, np.cumsum(cols_a >= 0) - 1, -1)
        cols_b = np.concatenate([np.arange(n) if o != "D" else np.full(n, -1) for o, n in ops])
        cols_b = np.where(cols_b >= 0, np.cumsum(cols_b >= 0) - 1, -1)
        ncol = len(cols_a)
        for run in range(1 + (k % 3 == 2)):
            c0 = int(rng.integers(1000, ncol - 1200)) if run == 0 else int(rng.integers(200, 700))
            n_n = int(rng.integers(100, 180)) if k % 4 else 99  # (99: one short of a cut)
            ia = cols_a[c0:c0 + n_n]
            a[ia[ia >= 0]] = 78
            if k % 3 != 1:
                ib = cols_b[c0:c0 + n_n]
                b[ib[ib >= 0]] = 78
        rcf = k % 2 == 1
        low = rng.random(len(a)) < 0.3
        a = np.where(low & (a != 78), a + 32, a).astype(np.uint8)
        b_out = np.array([comp[int(c)] for c in b[::-1]], np.uint8) if rcf else b
        qs = put(a)
        put(alpha[rng.integers(0, 4, int(rng.integers(50, 300)))])
        rs = put(b_out)
        put(alpha[rng.integers(0, 4, int(rng.integers(50, 300)))])
        cigar = "".join("%d%s" % (n, o) for o, n in ops)
        bed.append("chrH\t%d\t%d\tchrH\t%d\t%d\tx\t0\t+\t%s\t%d\t%d\t%s\tm=0;g=0" % (
            qs, qs + len(a), rs, rs + len(b), "-" if rcf else "+", max(len(a), len(b)), sum(n for _, n in ops), cigar))
    seq = np.concatenate(chunks).tobytes().decode()
    fa = str(tmp_path / "hand.fa")
    with open(fa, "w") as f:
        f.write(">chrH\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + "\n")
    with open(fa + ".fai", "w") as f:
        f.write("chrH\t%d\t6\t60\t61\n" % len(seq))
    bedp = str(tmp_path / "hand.bed")
    open(bedp, "w").write("\n".join(bed) + "\n")
    return fa, {"chrH": seq}, bedp


def test_stats_cuts_at_assembly_gaps_like_the_model(host, oracle, tmp_path):
    rng = np.random.default_rng(77)
    fa, genome, bed = _handmade(tmp_path, rng)
    out = str(tmp_path / "stats.tsv")
    lines, hits, pieces, columns = host.stats_generate(fa, bed, out, test_cols=_cols_hook(oracle))
    text = open(out).read().splitlines()
    exp = model_table(genome, open(bed).read().splitlines())
    assert text[1:] == exp
    assert hits == 12 and pieces > hits  # alignments were cut
    out2 = str(tmp_path / "stats2.tsv")
    host.stats_generate(fa, bed, out2, uppercase=10, max_error=0.9, max_ok_gap=0, min_split=200, test_cols=_cols_hook(oracle))
    assert open(out2).read().splitlines()[1:] == model_table(genome, open(bed).read().splitlines(), uppercase=10, max_error=0.9,
                                                             max_ok_gap=0, min_split=200)


@pytest.mark.gpu
def test_stats_generate_on_the_device_equals_oracle_columns(host, oracle, tmp_path):
    from sedef_amd.host import CLI
    fa, genome, bed = _stage(host, oracle, tmp_path, seed=22)
    cpu, gpu = str(tmp_path / "cpu.tsv"), str(tmp_path / "gpu.tsv")
    a = host.stats_generate(fa, bed, cpu, test_cols=_cols_hook(oracle))
    b = host.stats_generate(fa, bed, gpu)
    assert a == b and a[0] >= 10
    assert open(gpu).read() == open(cpu).read()
    r = subprocess.run([CLI, "stats", "generate", fa, bed], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(cpu).read() and "done!" in r.stderr
    r = subprocess.run([CLI, "stats", "generate", "--max-ok-gap", "1", "--min-split", "300", fa, bed], capture_output=True, text=True)
    host.stats_generate(fa, bed, cpu, max_ok_gap=1, min_split=300, test_cols=_cols_hook(oracle))
    assert r.returncode == 0 and r.stdout == open(cpu).read()
    fa2, genome2, bed2 = _handmade(tmp_path, np.random.default_rng(78))
    host.stats_generate(fa2, bed2, cpu, test_cols=_cols_hook(oracle))
    host.stats_generate(fa2, bed2, gpu)
    assert open(gpu).read() == open(cpu).read() and open(gpu).read().count("\n") >= 10
