"""ctypes bindings for the parity oracle.  TEST INFRASTRUCTURE ONLY.

`oracle.lib`  -> liboracle.so (our C restatement, pbwt_oracle.c)
`oracle.ref`  -> _ref/libpbwtref.so (the real reference compiled in place; may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package; the
product (pbwt_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    """compile liboracle.so (and _ref/libpbwtref.so when /root/reference is present)"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


class Match(C.Structure):
    _fields_ = [("ai", C.c_int32), ("bi", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class MatchVec(C.Structure):
    _fields_ = [("v", C.POINTER(Match)), ("n", C.c_size_t), ("cap", C.c_size_t)]


MATCH_DTYPE = np.dtype([("ai", "<i4"), ("bi", "<i4"), ("start", "<i4"), ("end", "<i4")])


def _p(arr, ctype):
    if arr is None:
        return None
    return arr.ctypes.data_as(C.POINTER(ctype))


def _load(path):
    return C.CDLL(path) if os.path.exists(path) else None


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_pack3.restype = C.c_size_t
        _lib.orc_unpack3.restype = C.c_size_t
        _lib.orc_checksum_i32.restype = C.c_uint64
        _lib.orc_checksum_u8.restype = C.c_uint64
    return _lib


def ref():
    """the real reference, or None when oracle/_ref was not built"""
    global _ref
    if _ref is None:
        # PBWT_ORACLE_REF_LIB: tests/test_integration.py points this at _ref/libpbwtref_gpu.so (the reference compiled with
        # integration/pbwtGpu.c, i.e. with the product underneath) to drive it through the same wrappers below
        path = os.environ.get("PBWT_ORACLE_REF_LIB") or os.path.join(_HERE, "_ref", "libpbwtref.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference"):
            build()
        _ref = _load(path)
        if _ref is not None:
            _ref.ref_build_bitcols.restype = C.c_long
            _ref.ref_max_within.restype = C.c_long
            _ref.ref_match_sweep.restype = C.c_long
            _ref.ref_pack3.restype = C.c_size_t
            _ref.ref_unpack3.restype = C.c_size_t
    return _ref


def wpc_for(M):
    """32-bit words per bit column (padded to 16 bytes)"""
    return ((M + 31) // 32 + 3) // 4 * 4


# ------------------------------------------------------------------ helpers over liboracle
def pack_bitcols(hap):
    """hap: uint8 [N, M] (site-major alleles, original order) -> uint32 [N, wpc]"""
    hap = np.ascontiguousarray(hap, dtype=np.uint8)
    N, M = hap.shape
    wpc = wpc_for(M)
    padded = np.zeros((N, wpc * 32), dtype=np.uint8)
    padded[:, :M] = hap
    return np.packbits(padded, axis=1, bitorder="little").view("<u4").reshape(N, wpc).copy()


def unpack_bitcols(bits, M):
    bits = np.ascontiguousarray(bits, dtype="<u4")
    return np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :M].copy()


def synth_bitcols(M, ncols, seed=1, kind=0, k0=0):
    wpc = wpc_for(M)
    bits = np.zeros((ncols, wpc), dtype=np.uint32)
    lib().orc_synth_bitcols(C.c_int(M), C.c_int(k0), C.c_int(ncols), C.c_int(wpc), C.c_uint64(seed),
                            C.c_int(kind), _p(bits, C.c_uint32))
    return bits


def pack3(y):
    y = np.ascontiguousarray(y, dtype=np.uint8)
    out = np.zeros(len(y) + 8, dtype=np.uint8)
    n = lib().orc_pack3(_p(y, C.c_uint8), C.c_int(len(y)), _p(out, C.c_uint8))
    return out[:n].copy()


def unpack3(z, M):
    z = np.ascontiguousarray(z, dtype=np.uint8)
    y = np.zeros(M, dtype=np.uint8)
    n0 = C.c_int(0)
    used = lib().orc_unpack3(_p(z, C.c_uint8), C.c_int(M), _p(y, C.c_uint8), C.byref(n0))
    return y, int(used), n0.value


def checksum_i32(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return int(lib().orc_checksum_i32(_p(v, C.c_int32), C.c_size_t(v.size)))


def checksum_u8(v):
    v = np.ascontiguousarray(v, dtype=np.uint8)
    return int(lib().orc_checksum_u8(_p(v, C.c_uint8), C.c_size_t(v.size)))


def build_bitcols(bits, M, with_d=True, k0=0, a0=None, d0=None, dump_sites=(), want_yz=True, want_csum=True):
    """the build loop.  Returns dict(yz, aFend, csum_a, csum_d, a_dump, d_dump, a_final, d_final)"""
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    ncols, wpc = bits.shape
    yz = np.zeros(ncols * M + 16, dtype=np.uint8) if want_yz else None
    nz = C.c_size_t(0)
    aFend = np.zeros(M, dtype=np.int32)
    ca = np.zeros(ncols + 1, dtype=np.uint64) if want_csum else None
    cd = np.zeros(ncols + 1, dtype=np.uint64) if want_csum else None
    ds = np.asarray(list(dump_sites), dtype=np.int32)
    a_dump = np.zeros((len(ds), M), dtype=np.int32)
    d_dump = np.zeros((len(ds), M + 1), dtype=np.int32)
    a_io = None if a0 is None else np.ascontiguousarray(a0, dtype=np.int32).copy()
    d_io = None if d0 is None else np.ascontiguousarray(d0, dtype=np.int32).copy()
    if a_io is None:
        a_io = np.arange(M, dtype=np.int32)
    if d_io is None:
        d_io = np.zeros(M + 1, dtype=np.int32)
        d_io[0] = d_io[M] = k0 + 1
    rc = lib().orc_build_bitcols(C.c_int(M), C.c_int(ncols), C.c_int(k0), _p(bits, C.c_uint32), C.c_int(wpc),
                                 C.c_int(1 if with_d else 0), _p(a_io, C.c_int32), _p(d_io, C.c_int32),
                                 _p(yz, C.c_uint8), C.c_size_t(0 if yz is None else yz.size), C.byref(nz),
                                 _p(aFend, C.c_int32), _p(ca, C.c_uint64), _p(cd, C.c_uint64),
                                 _p(ds, C.c_int32), C.c_int(len(ds)), _p(a_dump, C.c_int32), _p(d_dump, C.c_int32))
    assert rc == 0
    return dict(yz=None if yz is None else yz[:nz.value].copy(), aFend=aFend, csum_a=ca, csum_d=cd,
                a_dump=a_dump, d_dump=d_dump, a_final=a_io, d_final=d_io)


def segment(bits, M, k0, n_total, a0, d0, want_yz=True, want_hist=True, yz_cap=None):
    """a block of sites of build + -stats maxWithin continued from the cursor (a0, d0) at site k0 (orc_segment): returns
    dict(a, d: the cursor after the block; yz: the block's pack3 bytes; hist: int64[n_total + 2], the block's reports).
    Releases the GIL for the duration (ctypes): blocks can be checked from several threads at once."""
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    ncols, wpc = bits.shape
    a = np.ascontiguousarray(a0, dtype=np.int32).copy()
    d = np.ascontiguousarray(d0, dtype=np.int32).copy()
    assert a.size == M and d.size == M + 1
    yz = np.zeros((ncols * M + 16) if yz_cap is None else int(yz_cap), dtype=np.uint8) if want_yz else None   # (yz_cap: the bytes expected + M of slack)
    hist = np.zeros(n_total + 2, dtype=np.int64) if want_hist else None
    nz = C.c_size_t(0)
    rc = lib().orc_segment(C.c_int(M), C.c_int(ncols), C.c_int(k0), C.c_int(n_total), _p(bits, C.c_uint32), C.c_int(wpc),
                           _p(a, C.c_int32), _p(d, C.c_int32), _p(yz, C.c_uint8), C.c_size_t(0 if yz is None else yz.size), C.byref(nz),
                           _p(hist, C.c_int64), C.c_int(0 if hist is None else hist.size))
    assert rc == 0, rc
    return dict(a=a, d=d, yz=None if yz is None else yz[:nz.value].copy(), hist=hist)


def sweep_AD(yz, M, N, aFstart=None, dump_sites=()):
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    ca = np.zeros(N + 1, dtype=np.uint64)
    cd = np.zeros(N + 1, dtype=np.uint64)
    cy = np.zeros(N + 1, dtype=np.uint64)
    ds = np.asarray(list(dump_sites), dtype=np.int32)
    a_dump = np.zeros((len(ds), M), dtype=np.int32)
    d_dump = np.zeros((len(ds), M + 1), dtype=np.int32)
    y_dump = np.zeros((len(ds), M), dtype=np.uint8)
    c_dump = np.zeros(len(ds), dtype=np.int32)
    lib().orc_sweep_AD(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32),
                       _p(ca, C.c_uint64), _p(cd, C.c_uint64), _p(cy, C.c_uint64),
                       _p(ds, C.c_int32), C.c_int(len(ds)), _p(a_dump, C.c_int32), _p(d_dump, C.c_int32),
                       _p(y_dump, C.c_uint8), _p(c_dump, C.c_int32))
    return dict(csum_a=ca, csum_d=cd, csum_y=cy, a_dump=a_dump, d_dump=d_dump, y_dump=y_dump, c_dump=c_dump)


def _take(mv):
    out = np.zeros(mv.n, dtype=MATCH_DTYPE)
    if mv.n:
        C.memmove(out.ctypes.data, mv.v, mv.n * C.sizeof(Match))
    lib().orc_free(mv.v)
    return out


def max_within(yz, M, N, aFstart=None):
    """records in callback order (zero-length reports included)"""
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    mv = MatchVec()
    rc = lib().orc_max_within(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32),
                              C.c_int(0), C.byref(mv), None, C.c_int(0))
    assert rc == 0
    return _take(mv)


def max_within_range(yz, M, N, k_lo, k_hi, aFstart=None):
    """records of the sites k_lo <= k < k_hi only, in callback order"""
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    mv = MatchVec()
    rc = lib().orc_max_within_range(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32),
                                    C.c_int(k_lo), C.c_int(k_hi), C.byref(mv))
    assert rc == 0
    return _take(mv)


def max_within_hist(yz, M, N, aFstart=None):
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    hist = np.zeros(N + 2, dtype=np.int64)
    rc = lib().orc_max_within(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32),
                              C.c_int(1), None, _p(hist, C.c_int64), C.c_int(hist.size))
    assert rc == 0
    return hist


def long_within(yz, M, N, L, aFstart=None):
    """-longWithin L (matchLongWithin2): records in callback order"""
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    mv = MatchVec()
    rc = lib().orc_long_within(C.c_int(M), C.c_int(N), C.c_int(L), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32), C.byref(mv))
    assert rc == 0
    return _take(mv)


def match_sweep(pz, Mp, qz, Mq, N, pStart=None, qStart=None):
    pz = np.ascontiguousarray(pz, dtype=np.uint8)
    qz = np.ascontiguousarray(qz, dtype=np.uint8)
    pa = np.arange(Mp, dtype=np.int32) if pStart is None else np.ascontiguousarray(pStart, dtype=np.int32)
    qa = np.arange(Mq, dtype=np.int32) if qStart is None else np.ascontiguousarray(qStart, dtype=np.int32)
    mv = MatchVec()
    nomatch = C.c_int64(0)
    tot = (C.c_int64 * 2)()
    rc = lib().orc_match_sweep(C.c_int(Mp), C.c_int(N), _p(pz, C.c_uint8), C.c_size_t(pz.size), _p(pa, C.c_int32),
                               C.c_int(Mq), _p(qz, C.c_uint8), C.c_size_t(qz.size), _p(qa, C.c_int32),
                               C.byref(mv), C.byref(nomatch), tot)
    assert rc == 0
    return _take(mv), nomatch.value, (tot[0], tot[1])


MATCH5_DTYPE = np.dtype([("ai", "<i4"), ("bi", "<i4"), ("start", "<i4"), ("end", "<i4"), ("sparse", "<i4")])


class Match5Vec(C.Structure):
    _fields_ = [("v", C.c_void_p), ("n", C.c_size_t), ("cap", C.c_size_t)]


def match_sweep_sparse(pz, Mp, qz, Mq, N, nSparse, pStart=None, qStart=None):
    """matchSequencesSweepSparse (pbwtMatch.c:501-602): records (ai, bi, start, end, sparse) in callback order"""
    pz = np.ascontiguousarray(pz, dtype=np.uint8)
    qz = np.ascontiguousarray(qz, dtype=np.uint8)
    pa = np.arange(Mp, dtype=np.int32) if pStart is None else np.ascontiguousarray(pStart, dtype=np.int32)
    qa = np.arange(Mq, dtype=np.int32) if qStart is None else np.ascontiguousarray(qStart, dtype=np.int32)
    mv = Match5Vec()
    nomatch = C.c_int64(0)
    tot = (C.c_int64 * 2)()
    rc = lib().orc_match_sweep_sparse(C.c_int(Mp), C.c_int(N), _p(pz, C.c_uint8), C.c_size_t(pz.size), _p(pa, C.c_int32),
                                      C.c_int(Mq), _p(qz, C.c_uint8), C.c_size_t(qz.size), _p(qa, C.c_int32), C.c_int(nSparse),
                                      C.byref(mv), C.byref(nomatch), tot)
    assert rc == 0
    out = np.zeros(mv.n, dtype=MATCH5_DTYPE)
    if mv.n:
        C.memmove(out.ctypes.data, mv.v, mv.n * MATCH5_DTYPE.itemsize)
    lib().orc_free(C.c_void_p(mv.v))
    return out, nomatch.value, (tot[0], tot[1])


def ref_match_sweep_sparse(pz, Mp, qz, Mq, N, nSparse, pStart=None, qStart=None):
    r = ref()
    pz = np.ascontiguousarray(pz, dtype=np.uint8)
    qz = np.ascontiguousarray(qz, dtype=np.uint8)
    pa = np.arange(Mp, dtype=np.int32) if pStart is None else np.ascontiguousarray(pStart, dtype=np.int32)
    qa = np.arange(Mq, dtype=np.int32) if qStart is None else np.ascontiguousarray(qStart, dtype=np.int32)
    ptr = C.c_void_p()
    r.ref_match_sweep_sparse.restype = C.c_long
    n = r.ref_match_sweep_sparse(C.c_int(Mp), C.c_int(N), _p(pz, C.c_uint8), C.c_long(pz.size), _p(pa, C.c_int32),
                                 C.c_int(Mq), _p(qz, C.c_uint8), C.c_long(qz.size), _p(qa, C.c_int32), C.c_int(nSparse), C.byref(ptr))
    out = np.zeros(n, dtype=MATCH5_DTYPE)
    if n:
        C.memmove(out.ctypes.data, ptr, n * MATCH5_DTYPE.itemsize)
    C.CDLL(None).free(ptr)
    return out


def haplotypes(yz, M, N, aFstart=None):
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    out = np.zeros((N, M), dtype=np.uint8)
    lib().orc_haplotypes(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_size_t(yz.size), _p(a0, C.c_int32),
                         _p(out, C.c_uint8))
    return out


# ------------------------------------------------------------------ helpers over the real reference
def ref_build_bitcols(bits, M, with_d=True):
    r = ref()
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    N, wpc = bits.shape
    yz = np.zeros(N * M + 16, dtype=np.uint8)
    aFend = np.zeros(M, dtype=np.int32)
    a_all = np.zeros((N + 1, M), dtype=np.int32)
    d_all = np.zeros((N + 1, M + 1), dtype=np.int32)
    nz = r.ref_build_bitcols(C.c_int(M), C.c_int(N), _p(bits, C.c_uint32), C.c_int(wpc), C.c_int(1 if with_d else 0),
                             _p(yz, C.c_uint8), C.c_long(yz.size), _p(aFend, C.c_int32), _p(a_all, C.c_int32),
                             _p(d_all, C.c_int32))
    assert nz >= 0
    return dict(yz=yz[:nz].copy(), aFend=aFend, a_all=a_all, d_all=d_all)


def ref_time_build_and_within(bits, M):
    """the real reference timed on this host: (t_build_AD_pack3, t_maxWithin_stats) in seconds.
    NOTE: leaves the reference's histogram static set in this process (see ref_driver.c), so call it
    last or in a process that does not need ref_max_within afterwards."""
    r = ref()
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    N, wpc = bits.shape
    tb, tw = C.c_double(0), C.c_double(0)
    r.ref_time_build_and_within.restype = C.c_long
    r.ref_time_build_and_within(C.c_int(M), C.c_int(N), _p(bits, C.c_uint32), C.c_int(wpc), C.byref(tb), C.byref(tw))
    return tb.value, tw.value


def ref_sweep_dump(yz, M, N, aFstart=None):
    r = ref()
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    a_all = np.zeros((N + 1, M), dtype=np.int32)
    d_all = np.zeros((N + 1, M + 1), dtype=np.int32)
    y_all = np.zeros((N + 1, M), dtype=np.uint8)
    c_all = np.zeros(N + 1, dtype=np.int32)
    r.ref_sweep_dump(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_long(yz.size), _p(a0, C.c_int32),
                     _p(a_all, C.c_int32), _p(d_all, C.c_int32), _p(y_all, C.c_uint8), _p(c_all, C.c_int32))
    return dict(a_all=a_all, d_all=d_all, y_all=y_all, c_all=c_all)


def _ref_take(ptr, n):
    out = np.zeros(n, dtype=MATCH_DTYPE)
    if n:
        C.memmove(out.ctypes.data, ptr, n * C.sizeof(Match))
    ref().ref_free(ptr)
    return out


def ref_max_within(yz, M, N, aFstart=None):
    r = ref()
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    ptr = C.POINTER(Match)()
    n = r.ref_max_within(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_long(yz.size), _p(a0, C.c_int32), C.byref(ptr))
    return _ref_take(ptr, n)


def ref_max_within_file(yz, M, N, path, aFstart=None, hist=False, check=False):
    r = ref()
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    fn = r.ref_max_within_hist_to_file if hist else r.ref_max_within_text_to_file
    rc = fn(C.c_int(M), C.c_int(N), _p(yz, C.c_uint8), C.c_long(yz.size), _p(a0, C.c_int32),
            C.c_char_p(path.encode()), C.c_int(1 if check else 0))
    assert rc == 0


def ref_long_within_file(yz, M, N, L, path, aFstart=None, check=False):
    r = ref()
    yz = np.ascontiguousarray(yz, dtype=np.uint8)
    a0 = np.arange(M, dtype=np.int32) if aFstart is None else np.ascontiguousarray(aFstart, dtype=np.int32)
    rc = r.ref_long_within_text_to_file(C.c_int(M), C.c_int(N), C.c_int(L), _p(yz, C.c_uint8), C.c_long(yz.size), _p(a0, C.c_int32),
                                        C.c_char_p(path.encode()), C.c_int(1 if check else 0))
    assert rc == 0


def ref_match_sweep(pz, Mp, qz, Mq, N, pStart=None, qStart=None):
    r = ref()
    pz = np.ascontiguousarray(pz, dtype=np.uint8)
    qz = np.ascontiguousarray(qz, dtype=np.uint8)
    pa = np.arange(Mp, dtype=np.int32) if pStart is None else np.ascontiguousarray(pStart, dtype=np.int32)
    qa = np.arange(Mq, dtype=np.int32) if qStart is None else np.ascontiguousarray(qStart, dtype=np.int32)
    ptr = C.POINTER(Match)()
    n = r.ref_match_sweep(C.c_int(Mp), C.c_int(N), _p(pz, C.c_uint8), C.c_long(pz.size), _p(pa, C.c_int32),
                          C.c_int(Mq), _p(qz, C.c_uint8), C.c_long(qz.size), _p(qa, C.c_int32), C.byref(ptr))
    return _ref_take(ptr, n)
