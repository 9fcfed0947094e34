"""The library's configuration struct (include/sedef_hip.h: sdf_config; sedef_amd/csrc/sdf_config.hip): one place for every
tunable and test switch, filled from the SDF_* environment once per context or handed over by the caller, validated."""
import ctypes as C

import pytest


def test_defaults_names_and_dump():
    import sedef_amd
    cfg = sedef_amd.Config(from_env=False)
    d = cfg.as_dict()
    assert d["SDF_NO_PAIR"] == 0 and d["SDF_PIPELINE"] == 1 and d["SDF_STRIPE_MIN"] == 400 and d["SDF_LANE_MIN"] == 8192
    assert d["SDF_SPLIT_MIN"] == -1 and d["SDF_STRIPE_SPIN_CAP"] == 1 << 24 and d["SDF_WORKSPACE_GIB"] == 0
    text = sedef_amd.describe_config()
    for name in d:  # every field of the dump is described, with its field name and default
        assert name in text
    assert len(d) >= 38


def test_set_by_variable_or_field_name_and_validation():
    import sedef_amd
    cfg = sedef_amd.Config(from_env=False, SDF_NO_PAIR=1, strip_cols=4, SDF_WORKSPACE_GIB=1.5, SDF_LANE_PLAN="sort")
    d = cfg.as_dict()
    assert d["SDF_NO_PAIR"] == 1 and d["SDF_STRIP_COLS"] == 4 and d["SDF_WORKSPACE_GIB"] == 1.5 and d["SDF_LANE_PLAN"] == 1
    assert "(not the default)" in cfg.dump()
    for bad in (dict(SDF_NO_SUCH_THING=1), dict(SDF_NO_PAIR="yes"), dict(SDF_NO_PAIR=2), dict(SDF_STRIPE_MIN=5),
                dict(SDF_STRIP_COLS=-1), dict(SDF_SPLIT_DIV=1)):
        with pytest.raises(sedef_amd.SdfError):
            sedef_amd.Config(from_env=False, **bad)


def test_environment_is_read_once_per_context_and_a_typo_is_an_error(monkeypatch):
    import sedef_amd
    monkeypatch.setenv("SDF_MIXED_MIN", "2")
    monkeypatch.setenv("SDF_DEBUG_TIMING", "")  # (flags that used to mean "set at all")
    monkeypatch.setenv("SDF_LANE_PLAN", "")      # (set to nothing: not set)
    d = sedef_amd.Config().as_dict()
    assert d["SDF_MIXED_MIN"] == 2 and d["SDF_DEBUG_TIMING"] == 1 and d["SDF_LANE_PLAN"] == 0
    monkeypatch.setenv("SDF_STRIPE_MIN", "12")  # out of range: sdf_create refuses, with the variable named
    with pytest.raises(sedef_amd.SdfError, match="SDF_STRIPE_MIN"):
        sedef_amd.Config()
    lib = sedef_amd.load_library()
    assert not lib.sdf_create(0, 0)
    assert b"SDF_STRIPE_MIN" in lib.sdf_last_error(None)
    monkeypatch.delenv("SDF_STRIPE_MIN")
    # a struct the library did not initialise (size field) is refused before anything touches a device
    lib.sdf_create_cfg.restype = C.c_void_p
    lib.sdf_create_cfg.argtypes = [C.c_int, C.c_size_t, C.c_void_p]
    raw = C.create_string_buffer(512)
    assert not lib.sdf_create_cfg(0, 0, raw)
    assert b"size field" in lib.sdf_last_error(None)
