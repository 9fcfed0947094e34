"""Stage-scale parity (BASELINE configs[0] / configs[2] shape): a multi-chromosome chr1-sized synthetic genome, thousands
of planted duplications (fwd + rc, soft-masking, N runs, hits clamped at chromosome ends) through `align bucket` ->
`align generate`, the product path on the GPU against the same host code with the REFERENCE kernel as the DP
(oracle/_ref: ksw_extz2_sse behind the library's test hook; the scalar oracle where _ref is not built).

The sha256 of the concatenated stdout is committed (tests/golden/stage_scale.sha256): the CPU leg in the build container
and the GPU leg on the MI355X box must both reproduce it."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import hostgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHA_FILE = os.path.join(ROOT, "tests", "golden", "stage_scale.sha256")
SHA_CHR1 = os.path.join(ROOT, "tests", "golden", "stage_chr1.sha256")
NBUCKETS = 4


@pytest.fixture(scope="module")
def host():
    from sedef_amd import host as h
    from sedef_amd.build import build_library
    build_library()
    h.build_host()
    return h


def cpu_dp_hook(oracle):
    """The reference kernel where oracle/_ref is built (8x faster than the scalar oracle), else the scalar oracle."""
    from oracle.binding import build_reference
    so = build_reference()
    if so:
        lib = C.CDLL(so)
        if hasattr(lib, "ref_extz2_hook"):
            return C.cast(lib.ref_extz2_hook, C.c_void_p), lib
    return C.cast(oracle.lib.sdfo_extz2, C.c_void_p), oracle.lib


_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def check_bedpe(lines, genome):
    """Every output line is consistent with the genome: CIGAR spans == coordinate spans, span column, >= 900 columns,
    and the m=/g= error columns recomputed from the sequences (src/hit.cc:173-195, src/align.cc:274-315)."""
    up = {k: np.frombuffer(v.upper().encode(), dtype=np.uint8) for k, v in genome.items()}
    for ln in lines:
        f = ln.split("\t")
        assert len(f) == 28
        qs, qe, rs, re_ = int(f[1]), int(f[2]), int(f[4]), int(f[5])
        assert 0 <= qs < qe <= len(up[f[0]]) and 0 <= rs < re_ <= len(up[f[3]])
        ops = re.findall(r"(\d+)([MID])", f[12])
        assert "".join(n + o for n, o in ops) == f[12]
        lens = np.array([int(n) for n, _ in ops], np.int64)
        kind = np.array([o for _, o in ops])
        qadv = np.where(kind != "I", lens, 0)
        radv = np.where(kind != "D", lens, 0)
        assert qadv.sum() == qe - qs and radv.sum() == re_ - rs and int(f[11]) == lens.sum() >= 900
        a = up[f[0]][qs:qe]
        b = up[f[3]][rs:re_]
        if f[9] == "-":
            b = _COMP[b[::-1]]
        q0 = np.cumsum(qadv) - qadv
        r0 = np.cumsum(radv) - radv
        m = kind == "M"
        within = np.arange(lens[m].sum()) - np.repeat(np.cumsum(lens[m]) - lens[m], lens[m])
        qi = np.repeat(q0[m], lens[m]) + within
        ri = np.repeat(r0[m], lens[m]) + within
        ma = int(((a[qi] == b[ri]) & (a[qi] != 78)).sum())
        mm = int(len(qi) - ma)
        gb = int(lens[~m].sum())
        assert f[13].startswith("m=%.1f;g=%.1f" % (100.0 * mm / (ma + mm + gb), 100.0 * gb / (ma + mm + gb))), ln[:200]


def run_stage(host, tmp_path, tag, genome_fa, test_dp, **scoring):
    out = tmp_path / ("buckets_" + tag)
    out.mkdir()
    host.bucket(genome_fa + ".seeds.bed", NBUCKETS, str(out), genome_fa)
    files = sorted(os.listdir(out))
    assert files == ["bucket_%04d" % b for b in range(NBUCKETS)]
    text, pairs, tasks = "", 0, 0
    for fn in files:
        res = str(tmp_path / ("%s_%s.bed" % (tag, fn)))
        st = host.generate(genome_fa, str(out / fn), 11, res, test_dp=test_dp, **scoring)
        pairs += st[0]
        tasks += st[2]
        text += open(res).read()
    return text, pairs, tasks


@pytest.fixture(scope="module")
def big_genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("stage")
    fa = str(d / "genome.fa")
    genome, nseeds = hostgen.make_big_genome(fa)
    return fa, genome, nseeds


def _committed():
    return open(SHA_FILE).read().split()[0] if os.path.exists(SHA_FILE) else None


def _summary(text):
    lines = text.splitlines()
    rc = sum(1 for ln in lines if ln.split("\t")[9] == "-")
    chroms = {ln.split("\t")[0] for ln in lines} | {ln.split("\t")[3] for ln in lines}
    return lines, rc, chroms


def test_stage_scale_cpu_reference_kernel(host, oracle, big_genome, tmp_path):
    """CPU leg: host pipeline + the reference's own ksw_extz2_sse as the DP -> the committed stdout hash."""
    fa, genome, nseeds = big_genome
    assert nseeds >= 5000 and len(genome) >= 3 and sum(len(v) for v in genome.values()) >= 50_000_000
    hook, _keep = cpu_dp_hook(oracle)
    text, pairs, tasks = run_stage(host, tmp_path, "cpu", fa, hook)
    lines, rc, chroms = _summary(text)
    assert pairs >= 4000 and len(lines) >= 3000 and rc >= 1000 and len(chroms) >= 3
    check_bedpe(lines, genome)
    # duplications planted flush against a chromosome end: the bucket stage extends the hit beyond the end, the fetch
    # clamps it (src/fasta.cc:112-116) and, for rc hits, the clamped end enters the coordinate arithmetic
    # (src/align_main.cc:317-321); the alignments reach the last bases of the chromosome on both strands
    near_end = set()
    for ln in lines:
        f = ln.split("\t")
        for name, end in ((f[0], int(f[2])), (f[3], int(f[5]))):
            if len(genome[name]) - end <= 8:
                near_end.add((name, f[9]))
    assert len(near_end) >= 3 and {s_ for _, s_ in near_end} == {"+", "-"}
    over = 0
    for b in range(NBUCKETS):
        for ln in open(tmp_path / "buckets_cpu" / ("bucket_%04d" % b)):
            f = ln.split("\t")
            over += int(f[2]) > len(genome[f[0]]) or int(f[5]) > len(genome[f[3]])
    assert over >= 4
    sha = hashlib.sha256(text.encode()).hexdigest()
    if os.environ.get("SDF_WRITE_GOLDEN"):
        open(SHA_FILE, "w").write("%s  %d lines, %d pairs, %d DP tasks (tests/hostgen.py: make_big_genome seed 7)\n" % (
            sha, len(lines), pairs, tasks))
    assert sha == _committed()


@pytest.mark.gpu
def test_stage_scale_gpu_equals_cpu_and_committed_hash(host, oracle, big_genome, tmp_path):
    fa, genome, nseeds = big_genome
    hook, _keep = cpu_dp_hook(oracle)
    cpu, pairs, _ = run_stage(host, tmp_path, "cpu", fa, hook)
    gpu, gpairs, gtasks = run_stage(host, tmp_path, "gpu", fa, None)
    assert gpairs == pairs and gtasks > 100000
    assert gpu == cpu
    lines, rc, chroms = _summary(gpu)
    assert len(lines) >= 3000 and rc >= 1000 and len(chroms) >= 3
    check_bedpe(lines, genome)
    assert hashlib.sha256(gpu.encode()).hexdigest() == _committed()
    # the CLI itself on one bucket: same bytes on stdout
    from sedef_amd.host import CLI
    out = tmp_path / "buckets_cli"
    out.mkdir()
    r = subprocess.run([CLI, "align", "bucket", "-n", str(NBUCKETS), fa + ".seeds.bed", str(out), fa],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([CLI, "align", "generate", "-k", "11", fa, str(out / "bucket_0001")], capture_output=True,
                       text=True)
    assert r.returncode == 0 and "Finished" in r.stderr
    assert r.stdout == open(tmp_path / "gpu_bucket_0001.bed").read()
    # ... whatever the driver's settings: the seed anchors in one call or in three parts under the chaining of the part before
    # (round 6; by default a super-batch of this size takes them in one or two), the DP rounds on the resident characters or on
    # bases cut out on the host again, three lanes
    for env in ({"SDF_ANCHOR_PARTS": "1"}, {"SDF_ANCHOR_PARTS": "3"}, {"SDF_RESIDENT_DP": "0"},
                {"SDF_ANCHOR_PARTS": "5", "SDF_LANES": "3", "SDF_SUPER_BATCH": "256"}):
        r2 = subprocess.run([CLI, "align", "generate", "-k", "11", fa, str(out / "bucket_0001")], capture_output=True,
                            text=True, env={**os.environ, **env})
        assert r2.returncode == 0 and "Finished" in r2.stderr, (env, r2.stderr[-400:])
        assert r2.stdout == r.stdout, env


@pytest.mark.gpu
def test_stage_scoring_overrides_gpu_equals_cpu(host, oracle, tmp_path):
    """--match 3 --mismatch -5 --gap-open -20 --gap-extend -2 (src/align_main.cc:343-352) through the whole stage."""
    from sedef_amd.host import CLI
    fa = str(tmp_path / "genome.fa")
    genome, beds = hostgen.make_genome(fa, seed=11, glen=150000, nsd=16)
    hook, _keep = cpu_dp_hook(oracle)
    sc = dict(match=3, mismatch=-5, gap_open=-20, gap_extend=-2)
    cpu, gpu, dflt = (str(tmp_path / n) for n in ("cpu.bed", "gpu.bed", "default.bed"))
    host.generate(fa, fa + ".bed", 11, cpu, test_dp=hook, **sc)
    host.generate(fa, fa + ".bed", 11, gpu, **sc)
    host.generate(fa, fa + ".bed", 11, dflt)
    assert open(gpu).read() == open(cpu).read() and open(gpu).read().count("\n") >= 8
    assert open(gpu).read() != open(dflt).read()  # the overrides reach the DP
    r = subprocess.run([CLI, "align", "generate", "-k", "11", "--match", "3", "--mismatch", "-5", "--gap-open", "-20",
                        "--gap-extend", "-2", fa, fa + ".bed"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == open(cpu).read()


# ---------------------------------------------------------------- BASELINE configs[2] at size: chr1-vs-chr1, fwd + rc
@pytest.fixture(scope="module")
def chr1_genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("chr1")
    fa = str(d / "genome.fa")
    genome, nseeds = hostgen.make_chr1_genome(fa)
    return fa, genome, nseeds


def _chr1_checks(lines, genome, nseeds):
    """What the chr1-sized run must exercise besides byte equality: every line consistent with the genome, both strands,
    alignments of tens of kilobases, copies on the same chromosome right behind their source, chromosome ends."""
    assert nseeds >= 1800 and len(genome["chr1"]) >= 249_000_000
    assert len(lines) >= 1500
    check_bedpe(lines, genome)
    f = [ln.split("\t") for ln in lines]
    spans = np.array([int(x[11]) for x in f])
    assert (spans >= 50_000).sum() >= 40 and (spans >= 10_000).sum() >= 500 and spans.max() >= 90_000
    assert sum(1 for x in f if x[9] == "-") >= 500 and sum(1 for x in f if x[9] == "+") >= 500
    near = sum(1 for x in f if x[0] == x[3] and x[9] == "+" and abs(int(x[4]) - int(x[2])) < 3000)
    assert near >= 60  # tandem copies: query end and reference start a gap apart
    at_end = sum(1 for x in f if len(genome[x[3]]) - int(x[5]) <= 8 or int(x[4]) <= 8)
    assert at_end >= 2


def test_stage_chr1_scale_cpu_reference_kernel(host, oracle, chr1_genome, tmp_path):
    """configs[2] at size, CPU leg: a 249 Mb chromosome with 1-100 kb duplications at 2-25 % divergence through `align
    bucket` -> `align generate`, the reference's own ksw_extz2_sse as the DP -> the committed stdout hash."""
    fa, genome, nseeds = chr1_genome
    hook, _keep = cpu_dp_hook(oracle)
    text, pairs, tasks = run_stage(host, tmp_path, "cpu", fa, hook)
    lines, rc, chroms = _summary(text)
    assert nseeds - 20 <= pairs <= nseeds and tasks >= 500_000  # (`align bucket` merges seeds that overlap)
    _chr1_checks(lines, genome, nseeds)
    sha = hashlib.sha256(text.encode()).hexdigest()
    if os.environ.get("SDF_WRITE_GOLDEN"):
        open(SHA_CHR1, "w").write("%s  %d lines, %d pairs, %d DP tasks (tests/hostgen.py: make_chr1_genome seed 13)\n" % (
            sha, len(lines), pairs, tasks))
    assert sha == (open(SHA_CHR1).read().split()[0] if os.path.exists(SHA_CHR1) else None)


@pytest.mark.gpu
def test_stage_chr1_scale_gpu_equals_cpu_and_committed_hash(host, oracle, chr1_genome, tmp_path):
    fa, genome, nseeds = chr1_genome
    hook, _keep = cpu_dp_hook(oracle)
    cpu, pairs, _ = run_stage(host, tmp_path, "cpu", fa, hook)
    gpu, gpairs, gtasks = run_stage(host, tmp_path, "gpu", fa, None)
    assert gpairs == pairs and gtasks >= 500_000
    assert gpu == cpu
    lines, rc, chroms = _summary(gpu)
    _chr1_checks(lines, genome, nseeds)
    assert hashlib.sha256(gpu.encode()).hexdigest() == open(SHA_CHR1).read().split()[0]
