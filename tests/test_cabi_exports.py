"""CPU: the C-ABI library builds/loads and exports every symbol include/sedef_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sedef_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from sedef_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_host_helpers_without_gpu():
    import numpy as np

    import sedef_amd
    assert sedef_amd.packed_words(0) == 0
    assert sedef_amd.packed_words(16) == 2 and sedef_amd.packed_words(33) == 5
    w = sedef_amd.pack_codes(np.array([0, 1, 2, 3, 4, 3], np.uint8))
    assert w.tolist() == [3300, 16]
    assert sedef_amd.band_cells(1000, 1000, 128) == 240488


def test_band_cells_agrees_with_oracle(oracle):
    import sedef_amd
    for q, t, w in [(1, 1, -1), (10, 300, 5), (300, 10, 5), (1000, 977, 64), (5, 5, 0), (33, 64, 1)]:
        assert sedef_amd.band_cells(q, t, w) == oracle.band_cells(q, t, w)


def test_no_gpu_means_loud_failure():
    import sedef_amd
    lib = sedef_amd.load_library()
    if lib.sdf_device_count() == 0:
        import pytest
        with pytest.raises(sedef_amd.SdfError):
            sedef_amd.Extz2Engine(0)


def test_committed_pmc_counters_belong_to_the_current_kernel_source():
    """profiles/hbm_traffic.json (what bench.py prints as roofline.traffic / valu_issue) carries the sha256 of
    extz2_pair.hip as it was when the PMC passes were taken: a kernel change without a new profiles/collect.sh run would
    make the bench line say `traffic: null` -- caught here instead of at the end of a round."""
    import hashlib
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    with open(os.path.join(ROOT, "sedef_amd", "csrc", "extz2_pair.hip"), "rb") as f:
        assert d["kernel_source_sha256"] == hashlib.sha256(f.read()).hexdigest()
    assert 0.5 < d["bytes_per_step"] / 24.35e9 < 1.2  # (HBM bytes of the launch against its algorithmic bytes)
