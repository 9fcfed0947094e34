"""sedef_amd: MI355X-native implementation of SEDEF's `align` DP hot path.

The product is the C-ABI library built from sedef_amd/csrc (include/sedef_hip.h); this package
is the thin host-side mirror used by tests and bench.py.  There is no CPU fallback: importing
works anywhere, but creating an engine without the built library or without a HIP device raises.
"""
from .extz2 import (Extz2Engine, RESULT_DTYPE, TASK_DTYPE, ksw_extz2, library_path,  # noqa: F401
                    load_library, pack_codes, packed_words, band_cells, SdfError, Config, describe_config)
from .build import build_library  # noqa: F401
