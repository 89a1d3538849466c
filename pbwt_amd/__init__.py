"""pbwt_amd — MI355X (gfx950) engine for the PBWT hot path (richarddurbin/pbwt drop-in for
pbwtCursorForwardsA/AD, matchMaximalWithin, matchSequencesSweep and the pack3 codec).

The product is the C-ABI shared library `libpbwtgpu.so` (include/pbwt_amd.h); this package is a
thin ctypes binding used by the tests, bench.py and Python callers.  There is no CPU compute path:
importing works anywhere, but creating an Engine without a HIP device (or without the built
library) raises.
"""
from .api import Engine, PbwtAmdError, lib_path, load_library, wpc_for, pass_advance_many, MATCH_DTYPE  # noqa: F401
from .api import OPT_WITH_D, OPT_SORTED, OPT_WITHIN_HIST, OPT_CHECKSUM, OPT_PACK3, OPT_WITHIN_RECS  # noqa: F401
from .build import build_library, build_cli  # noqa: F401
