"""fast_align / chain_anchors' sweep / refine_chains / the stage driver against fixtures made with the REFERENCE's own classes.

tests/golden/stage_pairs_kat.json.gz (generator: tests/golden/make_golden_stage_pairs.py) holds what a definition-level
model of src/chain.cc:103-268, src/refine.cc:23-193 and src/align_main.cc:200-337 (tests/bruteforce.py: StageModel, control
flow only) produces when every alignment, merge, guide alignment, tree query, FASTA fetch and BED line on the way is made by
the reference's Alignment / Hit / SegmentTree / FastaReference classes compiled unmodified.  The product's host pipeline
(sedef_amd/csrc/host/pipeline.cc, chain.cc) has to reproduce it byte for byte: with the CPU test hook here, through the GPU
provider and the CLI under -m gpu."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from sedef_amd import host as h
    from sedef_amd.build import build_library
    build_library()
    h.build_host()
    return h


@pytest.fixture(scope="module")
def cpu_dp(oracle):
    """The DP behind the CPU test hook: the reference kernel itself when oracle/_ref is built (fast), else the oracle."""
    from oracle.binding import Reference
    try:
        ref = Reference()
        return C.cast(ref.lib.ref_extz2_hook, C.c_void_p)
    except Exception:
        return C.cast(oracle.lib.sdfo_extz2, C.c_void_p)


@pytest.fixture(scope="module")
def golden():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "stage_pairs_kat.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def _expect(c):
    return [tuple(e) for e in c["expect"]]


def _run_pair(host, c, **kw):
    return host.fast_align(c["query"], c["ref"], c["qname"], c["rname"], c["q_rc"], c["r_rc"], c["qstart"], c["rstart"],
                           c["kmer"], **kw)


def test_fixture_covers_the_cases_the_reference_distinguishes(golden):
    pairs, notes = golden["pairs"], golden["notes"]
    assert len(pairs) >= 200
    kinds = {p["kind"] for p in pairs}
    assert {"rc", "self_overlap", "tandem", "far_equal", "far_unequal", "far_cut", "threshold", "dp_tie"} <= kinds
    assert sum(bool(p["expect"]) for p in pairs) >= 150 and sum(not p["expect"] for p in pairs) >= 5
    # both sides of the 489.99999999999994 chain filter (src/chain.cc:233-238), ties of `sco >= dp[ai]` (src/refine.cc:93),
    # same-chromosome skips (src/refine.cc:42-53,80-88), merges (src/refine.cc:172), every drop of the path loop
    for k in ("threshold_drop", "threshold_keep", "refine_dp_tie", "refine_self_overlap_skip", "refine_self_overlap_gap",
              "refine_merge", "refine_guide_multi", "refine_est_size_drop", "refine_overlap_drop", "refine_final_size_drop"):
        assert notes.get(k, 0) >= 1, k
    assert any(any(k == "threshold_keep_490" for k in p["notes"]) for p in pairs)
    assert any(any(k == "threshold_drop_489" for k in p["notes"]) for p in pairs)
    # zero-length run of a far gap with equal sides stays between two M runs (src/align.cc:135)
    import re
    assert any(re.search(r"\d+M\d+M", e[4]) for p in pairs if p["kind"] == "far_equal" for e in p["expect"])
    assert len(golden["chains"]) >= 200 and sum(bool(c.get("ties")) for c in golden["chains"]) >= 20
    assert len(golden["stages"]) >= 3


def test_fast_align_pairs_match_reference_fixture(host, cpu_dp, golden):
    bad = []
    for i, c in enumerate(golden["pairs"]):
        if _run_pair(host, c, test_dp=cpu_dp) != _expect(c):
            bad.append((i, c["kind"]))
    assert not bad, bad


def test_chain_sweep_matches_reference_tree_fixture(host, oracle, golden):
    """chain_anchors (src/chain.cc:103-199): the host's flat-array tree and the oracle's restated tree against the sweep run
    on the reference's own SegmentTree -- tie winners included."""
    for c in golden["chains"]:
        a = np.array(c["anchors"], np.int32).reshape(-1, 4)
        path, bounds = host.chain_raw(a)
        assert path.tolist() == c["path"] and bounds.tolist() == c["bounds"]
        o = oracle.chain_anchors(a)
        assert o["path"].tolist() == c["path"] and o["bounds"].tolist() == c["bounds"]


def _write_stage(fx, d):
    fa = os.path.join(str(d), "g.fa")
    open(fa, "w").write(fx["fasta"])
    open(fa + ".fai", "w").write(fx["fai"])
    bed = os.path.join(str(d), "bucket_0000")
    open(bed, "w").write(fx["bed"])
    return fa, bed


def test_stage_output_matches_reference_fixture(host, cpu_dp, golden, tmp_path):
    """`sedef align generate`: ordering by complexity class, clamped ends, rc remap, both BED halves of every line."""
    for k, fx in enumerate(golden["stages"]):
        d = tmp_path / ("s%d" % k)
        d.mkdir()
        fa, bed = _write_stage(fx, d)
        out = str(d / "out.bed")
        host.generate(fa, bed, fx["kmer"], out, test_dp=cpu_dp)
        got = open(out).read().split("\n")
        assert got[-1] == "" and got[:-1] == fx["expect"], k


def test_model_live_against_the_host_pipeline(host, cpu_dp):
    """The same comparison on fresh seeded pairs when the reference classes are here (build container): the fixture is
    not the only place the model and the product meet."""
    from oracle.binding import ReferenceAlign
    try:
        ref = ReferenceAlign()
        ref.lib.ref_tab_open
    except Exception:
        pytest.skip("oracle/_ref/libref_align.so not built and /root/reference absent")
    import bruteforce
    import hostgen

    class Orig:
        pass
    rng = np.random.default_rng(99)
    for it in range(24):
        kind = [k for k in hostgen.STAGE_PAIR_KINDS if k != "long"][it % (len(hostgen.STAGE_PAIR_KINDS) - 1)]
        c = hostgen.stage_pair_case(rng, kind)
        o = Orig()
        o.qname, o.rname, o.q_rc, o.r_rc, o.qs, o.rs = c["qname"], c["rname"], c["q_rc"], c["r_rc"], c["qstart"], c["rstart"]
        m = bruteforce.StageModel(ref)
        try:
            ids = m.fast_align(c["query"], c["ref"], o, 11)
        except bruteforce.Ambiguous:
            continue
        exp = []
        for h in ids:
            g = ref.tab_get(h)
            exp.append((g["qs"], g["qe"], g["rs"], g["re"], ref.tab_cigar(h), g["matches"], g["mismatches"], g["gaps"],
                        g["gap_bases"]))
        c["kmer"] = 11
        assert _run_pair(host, c, test_dp=cpu_dp) == exp, (it, kind)


# ---------------------------------------------------------------- the product path: GPU provider, CLI
@pytest.mark.gpu
def test_fast_align_pairs_match_reference_fixture_gpu(host, golden):
    bad = []
    for i, c in enumerate(golden["pairs"]):
        if _run_pair(host, c) != _expect(c):
            bad.append((i, c["kind"]))
    assert not bad, bad


@pytest.mark.gpu
def test_stage_cli_matches_reference_fixture_gpu(host, golden, tmp_path):
    from sedef_amd.host import CLI
    for k, fx in enumerate(golden["stages"]):
        d = tmp_path / ("s%d" % k)
        d.mkdir()
        fa, bed = _write_stage(fx, d)
        want = "".join(line + "\n" for line in fx["expect"])
        for env_extra in ({}, {"SDF_LANES": "3", "SDF_SUPER_BATCH": "2"}):
            env = dict(os.environ, **env_extra)
            env.pop("SDF_DEVICES", None)
            r = subprocess.run([CLI, "align", "generate", "-k", str(fx["kmer"]), fa, bed], capture_output=True, text=True,
                               env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            assert r.stdout == want, (k, env_extra)
            assert "Finished" in r.stderr


@pytest.mark.gpu
def test_chain_sweep_matches_reference_tree_fixture_gpu(golden):
    import sedef_amd
    eng = sedef_amd.Extz2Engine(0)
    cases = [np.array(c["anchors"], np.int32).reshape(-1, 4) for c in golden["chains"]]
    got = eng.chain_batch(cases, 210, 4)
    for c, (gp, gb) in zip(golden["chains"], got):
        assert np.asarray(gp).tolist() == c["path"] and np.asarray(gb).tolist() == c["bounds"]
