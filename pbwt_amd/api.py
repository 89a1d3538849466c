"""ctypes binding of libpbwtgpu.so (include/pbwt_amd.h)."""
import ctypes as C
import os
import weakref
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

OPT_WITH_D, OPT_SORTED, OPT_WITHIN_HIST, OPT_CHECKSUM, OPT_PACK3, OPT_WITHIN_RECS = 1, 2, 4, 8, 16, 32
MATCH_DTYPE = np.dtype([("ai", "<i4"), ("bi", "<i4"), ("start", "<i4"), ("end", "<i4")])
MATCH5_DTYPE = np.dtype([("ai", "<i4"), ("bi", "<i4"), ("start", "<i4"), ("end", "<i4"), ("sparse", "<i4")])
REPORT5_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
REPORT_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_int, C.c_int)
SHARD_HANDLE_BYTES = 576              # PBWTAMD_SHARD_HANDLE_BYTES


class PbwtAmdError(RuntimeError):
    pass


def lib_path():
    # PBWTAMD_LIB: another build of the same C ABI (A/B measurements of two builds on one GPU box)
    return os.environ.get("PBWTAMD_LIB") or os.path.join(_HERE, "libpbwtgpu.so")


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64; two HIP runtimes in one
    process cannot both own the GPU.  When torch is installed, map ITS libamdhip64.so first: our
    library's NEEDED libamdhip64.so.7 then binds to that copy (same SONAME), and a later
    `import torch` re-uses the same file.  A plain C host without torch uses /opt/rocm's copy."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """load the HIP library; raises if it has not been built (no fallback)"""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise PbwtAmdError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
        _share_hip_runtime_with_torch()
        L = C.CDLL(p)
        L.pbwtamd_last_error.restype = C.c_char_p
        L.pbwtamd_engine_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pbwtamd_engine_destroy.argtypes = [C.c_void_p]
        for n in ("pbwtamd_engine_M", "pbwtamd_engine_wpc", "pbwtamd_engine_batch", "pbwtamd_sync"):
            getattr(L, n).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def pass_advance_many(engines, dptrs, ncols, ncols_avail, opts):
    """advance several engines of the same width, created on the SAME stream, with fused chain launches (pbwtamd_pass_advance_many):
    dptrs[p] = device address of panel p's bit columns"""
    L = load_library()
    P = len(engines)
    hs = (C.c_void_p * P)(*[e._h for e in engines])
    ps = (C.c_void_p * P)(*[C.c_void_p(int(d)) for d in dptrs])
    rc = L.pbwtamd_pass_advance_many(hs, C.c_int(P), ps, C.c_int(engines[0].wpc), C.c_int(ncols), C.c_int(ncols_avail), C.c_uint(opts))
    if rc:
        raise PbwtAmdError(L.pbwtamd_last_error().decode())


def wpc_for(M):
    return ((M + 31) // 32 + 3) // 4 * 4


def _take(L, rp, n, dtype):
    """records the library malloc'ed -> numpy array over the same memory (no copy: 10^7 records are 160-200 MB); freed with the last view"""
    if not n:
        L.pbwtamd_free(rp)
        return np.zeros(0, dtype)
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(rp.value)
    weakref.finalize(buf, L.pbwtamd_free, C.c_void_p(rp.value))
    return np.frombuffer(buf, dtype=dtype)


def _p(arr, ctype):
    return None if arr is None else arr.ctypes.data_as(C.POINTER(ctype))


def _i32(x, M=None):
    if x is None:
        return None
    return np.ascontiguousarray(x, dtype=np.int32)


class Engine:
    """device state for one panel of M haplotypes (the PbwtCursor of the reference, in HBM)"""

    def __init__(self, M, batch_sites=0, device=0, stream=None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.pbwtamd_engine_create(C.byref(h), int(device), int(M), int(batch_sites),
                                           C.c_void_p(stream) if stream else None)
        if rc:
            raise PbwtAmdError(self._L.pbwtamd_last_error().decode())
        self._h = h
        self.M = int(M)
        self.wpc = self._L.pbwtamd_engine_wpc(h)
        self.batch = self._L.pbwtamd_engine_batch(h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbwtamd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise PbwtAmdError(self._L.pbwtamd_last_error().decode())

    # ------------------------------------------------------------ host-buffer entry points
    def build(self, bitcols, with_d=False, aFstart=None, want_yz=True):
        bitcols = np.ascontiguousarray(bitcols, dtype=np.uint32)
        N, wpc = bitcols.shape
        aF = _i32(aFstart)
        aFend = np.zeros(self.M, np.int32)
        dFend = np.zeros(self.M + 1, np.int32)
        yzp = C.POINTER(C.c_uint8)()
        nz = C.c_int64(0)
        self._chk(self._L.pbwtamd_build(self._h, _p(bitcols, C.c_uint32), C.c_int(wpc), C.c_int(N), C.c_int(1 if with_d else 0),
                                        _p(aF, C.c_int32), C.byref(yzp) if want_yz else None, C.byref(nz),
                                        _p(aFend, C.c_int32), _p(dFend, C.c_int32)))
        yz = None
        if want_yz:
            yz = _take(self._L, C.cast(yzp, C.c_void_p), nz.value, np.dtype(np.uint8))     # (no copy: the library's buffer, freed with the last view)
        return dict(yz=yz, aFend=aFend, dFend=dFend if with_d else None)

    def sweep_AD(self, yz, N, aFstart=None, dump_sites=(), checksums=True):
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        ca = np.zeros(N + 1, np.uint64) if checksums else None
        cd = np.zeros(N + 1, np.uint64) if checksums else None
        cy = np.zeros(N + 1, np.uint64) if checksums else None
        ds = np.asarray(list(dump_sites), dtype=np.int32)
        a_dump = np.zeros((len(ds), self.M), np.int32)
        d_dump = np.zeros((len(ds), self.M + 1), np.int32)
        y_dump = np.zeros((len(ds), self.M), np.uint8)
        self._chk(self._L.pbwtamd_sweep_AD(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                           _p(ca, C.c_uint64), _p(cd, C.c_uint64), _p(cy, C.c_uint64),
                                           _p(ds, C.c_int32), C.c_int(len(ds)), _p(a_dump, C.c_int32), _p(d_dump, C.c_int32),
                                           _p(y_dump, C.c_uint8)))
        return dict(csum_a=ca, csum_d=cd, csum_y=cy, a_dump=a_dump, d_dump=d_dump, y_dump=y_dump)

    def cursor_at(self, yz, N, k, aFstart=None):
        """the PbwtCursor fields (a, d, y, c, u, nBlockStart, n) before site k of a packed panel"""
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        a = np.zeros(self.M, np.int32); d = np.zeros(self.M + 1, np.int32); y = np.zeros(self.M, np.uint8)
        u = np.zeros(self.M + 1, np.int32); c = C.c_int32(0); nbs = C.c_int64(0); n = C.c_int64(0)
        self._chk(self._L.pbwtamd_cursor_at(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32), C.c_int(k),
                                            _p(a, C.c_int32), _p(d, C.c_int32), _p(y, C.c_uint8), C.byref(c), _p(u, C.c_int32),
                                            C.byref(nbs), C.byref(n)))
        return dict(a=a, d=d, y=y, c=c.value, u=u, nBlockStart=nbs.value, n=n.value)

    def haplotypes(self, yz, N, aFstart=None):
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        out = np.zeros((N, self.M), np.uint8)
        self._chk(self._L.pbwtamd_haplotypes(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32), _p(out, C.c_uint8)))
        return out

    def max_within(self, yz, N, aFstart=None, mode="records", callback=None):
        """mode 'records' -> structured array in callback order; 'hist' -> int64[N+1];
        'callback' -> calls callback(ai,bi,start,end) per report"""
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        if mode == "hist":
            hist = np.zeros(N + 1, np.int64)
            self._chk(self._L.pbwtamd_max_within(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                                 None, None, None, _p(hist, C.c_int64), C.c_int(hist.size)))
            return hist
        if mode == "callback":
            fn = REPORT_FN(callback)
            self._chk(self._L.pbwtamd_max_within(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                                 fn, None, None, None, C.c_int(0)))
            return None
        rp = C.c_void_p()
        n = C.c_int64(0)
        self._chk(self._L.pbwtamd_max_within(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                             None, C.byref(rp), C.byref(n), None, C.c_int(0)))
        return _take(self._L, rp, n.value, MATCH_DTYPE)

    def max_within_range(self, yz, N, k_lo, k_hi, aFstart=None):
        """records (callback order) of the sites k_lo <= k < k_hi only"""
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        rp = C.c_void_p()
        n = C.c_int64(0)
        self._chk(self._L.pbwtamd_max_within_range(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                                   C.c_int(k_lo), C.c_int(k_hi), None, C.byref(rp), C.byref(n)))
        return _take(self._L, rp, n.value, MATCH_DTYPE)

    def match_sweep(self, pz, N, qz, Mq, pStart=None, qStart=None, callback=None):
        """matchSequencesSweep of a packed query panel against this engine's packed panel.
        Returns (records in callback order, n_nomatch, (nTot, totLen)); with callback, records is None"""
        pz = np.ascontiguousarray(pz, dtype=np.uint8)
        qz = np.ascontiguousarray(qz, dtype=np.uint8)
        pa, qa = _i32(pStart), _i32(qStart)
        rp = C.c_void_p()
        n = C.c_int64(0)
        nom = C.c_int64(0)
        tot = (C.c_int64 * 2)()
        fn = REPORT_FN(callback) if callback else None
        self._chk(self._L.pbwtamd_match_sweep(self._h, _p(pz, C.c_uint8), C.c_int64(pz.size), C.c_int(N), _p(pa, C.c_int32),
                                              C.c_int(Mq), _p(qz, C.c_uint8), C.c_int64(qz.size), _p(qa, C.c_int32),
                                              fn, None if callback else C.byref(rp), C.byref(n), C.byref(nom), tot))
        out = None
        if not callback:
            out = _take(self._L, rp, n.value, MATCH_DTYPE)
        return out, nom.value, (tot[0], tot[1])

    def match_sweep_stream(self, N, Mq, cols, on_records=None, pStart=None, qStart=None, panel_opts=0):
        """matchSequencesSweep with both panels STREAMED from the device (pbwtamd_match_sweep_stream): cols(site0, ncols) -> (panel_ptr, query_ptr), device
        addresses of `ncols` original-order bit columns of each panel from site0 on (valid until the next call); on_records(array of MATCH5_DTYPE) is
        called per batch (the array is valid during the call) — None collects the records.  panel_opts: OPT_WITHIN_HIST | OPT_PACK3 | OPT_CHECKSUM feed
        those consumers from the same pass.  Returns (records or None, n_nomatch, (nTot, totLen))"""
        pa, qa = _i32(pStart), _i32(qStart)
        kept, err = [], []

        def _cols(user, site0, ncols, pp, qq):
            try:
                a, b = cols(int(site0), int(ncols))
                pp[0] = C.c_void_p(int(a)); qq[0] = C.c_void_p(int(b))
                return 0
            except Exception as ex:                         # (never let an exception cross the C frame)
                err.append(ex)
                return 1

        def _recs(user, rp, n):
            try:
                arr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int32)), shape=(int(n) * 5,)).view(MATCH5_DTYPE)
                if on_records is None:
                    kept.append(arr.copy())
                else:
                    on_records(arr)
                return 0
            except Exception as ex:
                err.append(ex)
                return 1

        COLS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
        RECS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
        nom = C.c_int64(0)
        tot = (C.c_int64 * 2)()
        cf, rf = COLS_FN(_cols), RECS_FN(_recs)
        rc = self._L.pbwtamd_match_sweep_stream(self._h, C.c_int(N), _p(pa, C.c_int32), C.c_int(Mq), _p(qa, C.c_int32), cf, rf, None, C.c_uint(panel_opts),
                                                C.byref(nom), tot)
        if err:
            raise err[0]
        self._chk(rc)
        out = None
        if on_records is None:
            out = np.concatenate(kept) if kept else np.zeros(0, MATCH5_DTYPE)
        return out, nom.value, (tot[0], tot[1])

    def drain_packed(self, buf=None):
        """the pack3 bytes written since pass_begin / the previous drain, out of the engine (pbwtamd_drain_packed): into `buf` (uint8 array, reused by the
        caller) or a fresh array; returns the array view of the bytes"""
        n = C.c_int64(0)
        self._chk(self._L.pbwtamd_drain_packed(self._h, None, C.c_int64(0), C.byref(n)))
        if buf is None or buf.size < n.value:
            buf = np.empty(max(n.value, 1), np.uint8)
        self._chk(self._L.pbwtamd_drain_packed(self._h, _p(buf, C.c_uint8), C.c_int64(buf.size), C.byref(n)))
        return buf[: n.value]

    def set_query_range(self, lo, hi=0):
        """query sweeps process the queries lo <= jj < hi only and tag records with the query's PBWT rank (sparse >> 1); lo < 0: all"""
        self._chk(self._L.pbwtamd_set_query_range(self._h, C.c_int(lo), C.c_int(hi)))

    def nomatch_events(self):
        """(jj, x, k, isSparse) rows of the last query sweep's "no match to query" events, in the reference's log order"""
        ev = C.POINTER(C.c_int32)(); n = C.c_int64(0)
        self._chk(self._L.pbwtamd_get_nomatch_events(self._h, C.byref(ev), C.byref(n)))
        out = np.ctypeslib.as_array(ev, shape=(max(n.value, 1) * 4,))[: n.value * 4].reshape(-1, 4).copy()
        self._L.pbwtamd_free(ev)
        return out

    def match_sweep_sparse(self, pz, N, qz, Mq, nSparse, pStart=None, qStart=None, callback=None):
        """matchSequencesSweepSparse: as match_sweep plus nSparse sparse cursors; records carry `sparse`"""
        pz = np.ascontiguousarray(pz, dtype=np.uint8)
        qz = np.ascontiguousarray(qz, dtype=np.uint8)
        pa, qa = _i32(pStart), _i32(qStart)
        rp = C.c_void_p()
        n = C.c_int64(0)
        nom = C.c_int64(0)
        tot = (C.c_int64 * 2)()
        fn = REPORT5_FN(callback) if callback else None
        self._chk(self._L.pbwtamd_match_sweep_sparse(self._h, _p(pz, C.c_uint8), C.c_int64(pz.size), C.c_int(N), _p(pa, C.c_int32),
                                                     C.c_int(Mq), _p(qz, C.c_uint8), C.c_int64(qz.size), _p(qa, C.c_int32), C.c_int(nSparse),
                                                     fn, None if callback else C.byref(rp), C.byref(n), C.byref(nom), tot))
        out = None
        if not callback:
            out = _take(self._L, rp, n.value, MATCH5_DTYPE)
        return out, nom.value, (tot[0], tot[1])

    def long_within(self, yz, N, L, aFstart=None, callback=None):
        """-longWithin L: records in callback order (or calls callback per report)"""
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        aF = _i32(aFstart)
        rp = C.c_void_p()
        n = C.c_int64(0)
        fn = REPORT_FN(callback) if callback else None
        self._chk(self._L.pbwtamd_long_within(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32), C.c_int(L),
                                              fn, None if callback else C.byref(rp), C.byref(n)))
        if callback:
            return None
        return _take(self._L, rp, n.value, MATCH_DTYPE)

    def regather(self, yz, N, site_order=None, hap_select=None, aFstart=None, aStart_out=None, want_fwd_end=False):
        """panel transform on the device: returns dict(yz, aFend[, aFend_fwd]) of the new panel"""
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        so, hs, aF, aS = _i32(site_order), _i32(hap_select), _i32(aFstart), _i32(aStart_out)
        n_out = N if so is None else so.size
        M_out = self.M if hs is None else hs.size
        aFend = np.zeros(M_out, np.int32)
        fwd = np.zeros(self.M, np.int32) if want_fwd_end else None
        yzp = C.POINTER(C.c_uint8)(); nz = C.c_int64(0)
        self._chk(self._L.pbwtamd_regather(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(aF, C.c_int32),
                                           _p(so, C.c_int32), C.c_int(n_out), _p(hs, C.c_int32), C.c_int(M_out), _p(aS, C.c_int32),
                                           C.byref(yzp), C.byref(nz), _p(aFend, C.c_int32), _p(fwd, C.c_int32)))
        out = np.ctypeslib.as_array(yzp, shape=(max(nz.value, 1),))[: nz.value].copy()
        self._L.pbwtamd_free(yzp)
        return dict(yz=out, aFend=aFend, aFend_fwd=fwd)

    def pack3(self, sorted_bitcols):
        sb = np.ascontiguousarray(sorted_bitcols, dtype=np.uint32)
        N, wpc = sb.shape
        yzp = C.POINTER(C.c_uint8)()
        nz = C.c_int64(0)
        self._chk(self._L.pbwtamd_pack3(self._h, _p(sb, C.c_uint32), C.c_int(wpc), C.c_int(N), C.byref(yzp), C.byref(nz)))
        yz = np.ctypeslib.as_array(yzp, shape=(max(nz.value, 1),))[:nz.value].copy()
        self._L.pbwtamd_free(yzp)
        return yz

    def unpack3(self, yz, N):
        yz = np.ascontiguousarray(yz, dtype=np.uint8)
        out = np.zeros((N, self.wpc), np.uint32)
        self._chk(self._L.pbwtamd_unpack3(self._h, _p(yz, C.c_uint8), C.c_int64(yz.size), C.c_int(N), _p(out, C.c_uint32), C.c_int(self.wpc)))
        return out

    # ------------------------------------------------------------ device-buffer entry points
    def synth_device(self, dptr, k0, ncols, seed=1, kind=0):
        self._chk(self._L.pbwtamd_synth_device(self._h, C.c_void_p(dptr), C.c_int(k0), C.c_int(ncols), C.c_uint64(seed), C.c_int(kind)))

    def pass_begin(self, n_total, k0=0, aInit=None):
        a = _i32(aInit)
        self._chk(self._L.pbwtamd_pass_begin(self._h, _p(a, C.c_int32), C.c_int(k0), C.c_int(n_total)))

    def pass_advance(self, dptr, ncols, ncols_avail, opts):
        self._chk(self._L.pbwtamd_pass_advance(self._h, C.c_void_p(dptr), C.c_int(self.wpc), C.c_int(ncols), C.c_int(ncols_avail), C.c_uint(opts)))

    def pass_stop(self):
        self._chk(self._L.pbwtamd_pass_stop(self._h))

    def pass_set_d(self, d):
        d = np.ascontiguousarray(d, dtype=np.int32)
        assert d.size == self.M + 1
        self._chk(self._L.pbwtamd_pass_set_d(self._h, _p(d, C.c_int32)))

    def pass_end(self, opts):
        self._chk(self._L.pbwtamd_pass_end(self._h, C.c_uint(opts)))

    def sync(self):
        self._chk(self._L.pbwtamd_sync(self._h))

    def get_state(self, with_d=True):
        a = np.zeros(self.M, np.int32)
        d = np.zeros(self.M + 1, np.int32) if with_d else None
        self._chk(self._L.pbwtamd_get_state(self._h, _p(a, C.c_int32), _p(d, C.c_int32)))
        return a, d

    def get_hist(self, n):
        h = np.zeros(n, np.int64)
        self._chk(self._L.pbwtamd_get_hist(self._h, _p(h, C.c_int64), C.c_int(n)))
        return h

    def get_packed(self):
        """pack3 bytes written since pass_begin (OPT_PACK3)"""
        yz = C.POINTER(C.c_uint8)(); nz = C.c_int64(0)
        self._chk(self._L.pbwtamd_get_packed(self._h, C.byref(yz), C.byref(nz)))
        out = np.ctypeslib.as_array(yz, shape=(max(nz.value, 1),))[: nz.value].copy()
        self._L.pbwtamd_free(yz)
        return out

    def get_checksums(self, k_first, n):
        ca = np.zeros(n, np.uint64); cd = np.zeros(n, np.uint64); cy = np.zeros(n, np.uint64)
        self._chk(self._L.pbwtamd_get_checksums(self._h, C.c_int(k_first), C.c_int(n), _p(ca, C.c_uint64), _p(cd, C.c_uint64), _p(cy, C.c_uint64)))
        return ca, cd, cy

    def phase_profile(self, ntiles=1024):
        out = np.zeros((ntiles, 8), np.int64)
        n = self._L.pbwtamd_get_phase_profile(self._h, _p(out, C.c_int64), C.c_int(ntiles))
        if n < 0:
            raise PbwtAmdError(self._L.pbwtamd_last_error().decode())
        return out[:n]

    # ------------------------------------------------------------ position sharding (one rank of a panel; pbwt_amd/posshard.py)
    def shard_init(self, rank, world):
        """make this engine rank `rank` of `world`; returns the handle blob (bytes) the ranks all-gather"""
        buf = C.create_string_buffer(SHARD_HANDLE_BYTES)
        self._chk(self._L.pbwtamd_shard_init(self._h, C.c_int(rank), C.c_int(world), buf))
        return buf.raw

    def shard_connect(self, blobs):
        """blobs: the ranks' handle blobs in rank order"""
        allb = b"".join(blobs)
        self._chk(self._L.pbwtamd_shard_connect(self._h, C.c_char_p(allb)))

    def shard_range(self, rank):
        lo, hi = C.c_int(0), C.c_int(0)
        self._chk(self._L.pbwtamd_shard_range(self._h, C.c_int(rank), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def shard_stats(self):
        """{row_wait_us, row_waits, barrier_us, barriers}: what this rank's sharded chain spent waiting for its peers (pbwtamd_shard_stats)"""
        out = (C.c_uint64 * 4)()
        self._chk(self._L.pbwtamd_shard_stats(self._h, out))
        return {"row_wait_us": out[0] * 0.01, "row_waits": int(out[1]), "barrier_us": out[2] * 0.01, "barriers": int(out[3])}

    def shard_blocks(self):
        """(site0[], nsites[], byte_end[]) of the pack3 blocks this rank wrote since pass_begin"""
        n = C.c_int(0)
        self._chk(self._L.pbwtamd_shard_blocks(self._h, None, None, None, C.c_int(0), C.byref(n)))
        s0 = np.zeros(max(n.value, 1), np.int64); ns = np.zeros(max(n.value, 1), np.int64); be = np.zeros(max(n.value, 1), np.int64)
        self._chk(self._L.pbwtamd_shard_blocks(self._h, _p(s0, C.c_int64), _p(ns, C.c_int64), _p(be, C.c_int64), C.c_int(s0.size), C.byref(n)))
        return s0[: n.value], ns[: n.value], be[: n.value]

    def chain_timing(self):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._chk(self._L.pbwtamd_get_chain_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def chain_sites(self):
        n = C.c_int64(0)
        self._chk(self._L.pbwtamd_get_chain_sites(self._h, C.byref(n)))
        return n.value
