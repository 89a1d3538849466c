"""CPU: the C-ABI library builds, loads and exports every symbol include/pbwt_amd.h declares
(no compute calls: there is no GPU here), and refuses to run without a device."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pbwt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pbwtamd_[a-z0-9_]+)\s*\(", hdr)) - {"pbwtamd_report_fn"})


def test_library_exports_every_declared_symbol():
    import pbwt_amd
    pbwt_amd.build_library()
    L = pbwt_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libpbwtgpu.so does not export %s" % s
    assert L.pbwtamd_abi_version() == 4


def test_no_cpu_fallback():
    """without a HIP device the product must fail loudly, never compute on the CPU"""
    import pbwt_amd
    L = pbwt_amd.load_library()
    if L.pbwtamd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pbwt_amd.PbwtAmdError, match="no HIP device"):
        pbwt_amd.Engine(100)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under pbwt_amd/ or include/ may reference it"""
    for base in ("pbwt_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".inc", ".c", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "import oracle" not in txt and "liboracle" not in txt and "orc_" not in txt, f


def test_records_are_wrapped_not_copied():
    """record blocks the library malloc's are handed to numpy as they are (pbwt_amd.api._take): the array aliases the block, views keep
    it alive, and an empty result still frees it"""
    import ctypes as C
    import gc
    import numpy as np
    import pbwt_amd
    from pbwt_amd import api
    L = pbwt_amd.load_library()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    n = 5
    p = libc.malloc(n * api.MATCH_DTYPE.itemsize)
    raw = (C.c_int32 * (4 * n)).from_address(p)
    for i in range(4 * n):
        raw[i] = 100 + i
    arr = api._take(L, C.c_void_p(p), n, api.MATCH_DTYPE)
    assert arr.shape == (n,) and arr.ctypes.data == p                  # the same memory
    assert [int(x) for x in arr[2]] == [108, 109, 110, 111]
    tail = arr[3:]
    del arr
    gc.collect()
    assert [int(x) for x in tail[0]] == [112, 113, 114, 115]           # a view keeps the block alive
    empty = api._take(L, C.c_void_p(libc.malloc(16)), 0, api.MATCH_DTYPE)
    assert empty.shape == (0,)
