"""Run in a FRESH process by tests/test_integration.py (GPU box): drives oracle/_ref/libpbwtref_gpu.so — the reference's own
sources compiled with integration/pbwtMatchGpu.c in place of pbwtMatch.c — so the reference's OWN pbwtLongMatches,
reportMatch, -check and matchSequencesDynamic run with libpbwtgpu.so underneath, and diffs their output against the
goldens the CPU reference produced (tests/golden/).  A fresh process because some entry points fork (HIP must not be
initialised in the parent)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(HERE, "golden")


def write_pbwt(path, M, N, yz, aFstart=None, aFend=None):
    """.pbwt v3 (pbwtIO.c:33-57)"""
    a0 = np.arange(M, dtype="<i4") if aFstart is None else np.asarray(aFstart, "<i4")
    a1 = np.arange(M, dtype="<i4") if aFend is None else np.asarray(aFend, "<i4")
    with open(path, "wb") as f:
        f.write(b"PBW3"); f.write(np.array([M, N], "<i4").tobytes()); f.write(a0.tobytes()); f.write(a1.tobytes())
        f.write(np.array([len(yz)], "<i8").tobytes()); f.write(b"    "); f.write(np.asarray(yz, np.uint8).tobytes())


def match_text(recs):
    return "".join("MATCH\t%d\t%d\t%d\t%d\t%d\n" % (r["ai"], r["bi"], r["start"], r["end"], r["end"] - r["start"])
                   for r in recs if r["start"] != r["end"])


def main():
    import pbwt_amd
    pbwt_amd.load_library()                       # maps the HIP runtime the product uses; no device call yet
    lib = os.path.join(ROOT, "oracle", "_ref", "libpbwtref_gpu.so")
    os.environ["PBWT_ORACLE_REF_LIB"] = lib
    import oracle
    r = oracle.ref()
    assert r is not None and r.refgpu_is_gpu_build() == 1
    nchk = 0
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.txt")
        # ---- -stats first: ref_max_within_hist_to_file forks (the reference keeps its histogram static set afterwards), and
        # a child can only bring up HIP if the parent has not yet.  The histogram travels through pbwtMatch.c's
        # file-static matchLengthHist, filled by the replacement matchMaximalWithin in the same (unity) TU.
        for name in ("mosaic_M300_N400_k0.npz", "mosaic_M70_N150_k1.npz", "mosaic_M1100_N260_k0.npz"):
            g = np.load(os.path.join(GOLDEN, name))
            oracle.ref_max_within_file(g["yz"], int(g["M"]), int(g["N"]), out, hist=True, check=False)
            assert open(out).read() == bytes(g["hist_txt"]).decode(); nchk += 1
        # ---- the reference's own test panel: pbwtLongMatches + reportMatch + -check, GPU underneath
        g = np.load(os.path.join(GOLDEN, "merge1.npz"))
        M, N = int(g["M"]), int(g["N"])
        oracle.ref_max_within_file(g["yz"], M, N, out, aFstart=g["aFstart"], check=True)
        assert open(out).read() == open(os.path.join(GOLDEN, "merge1.maxwithin.txt")).read(); nchk += 1
        assert np.array_equal(oracle.ref_max_within(g["yz"], M, N, aFstart=g["aFstart"]), g["within"]); nchk += 1
        for name in ("mosaic_M300_N400_k0.npz", "mosaic_M70_N150_k1.npz", "mosaic_M1100_N260_k0.npz"):
            g = np.load(os.path.join(GOLDEN, name))
            M, N, Mq = int(g["M"]), int(g["N"]), int(g["Mq"])
            # callback stream of matchMaximalWithin as the reference's callers see it
            assert np.array_equal(oracle.ref_max_within(g["yz"], M, N), g["within"]); nchk += 1
            # -matchDynamic: the reference's matchSequencesDynamic (pbwtRead of the query file + reportMatch)
            pp, qp = os.path.join(td, "p.pbwt"), os.path.join(td, "q.pbwt")
            write_pbwt(pp, M - Mq, N, g["pz"]); write_pbwt(qp, Mq, N, g["qz"])
            assert r.refgpu_match_dynamic_to_file(pp.encode(), qp.encode(), out.encode()) == 0
            assert open(out).read() == match_text(g["qrecs"]); nchk += 1
            assert np.array_equal(oracle.ref_match_sweep(g["pz"], M - Mq, g["qz"], Mq, N), g["qrecs"]); nchk += 1
            # the build loop of pbwtReadMacs through pbwtBuildFromBitColumns
            bits = np.ascontiguousarray(g["bits"], np.uint32)
            yz = np.zeros(N * M + 16, np.uint8); aFend = np.zeros(M, np.int32)
            r.refgpu_build_bitcols.restype = C.c_long
            nz = r.refgpu_build_bitcols(C.c_int(M), C.c_int(N), bits.ctypes.data_as(C.c_void_p), C.c_int(bits.shape[1]),
                                        yz.ctypes.data_as(C.c_void_p), C.c_long(yz.size), aFend.ctypes.data_as(C.c_void_p))
            assert nz == len(g["yz"]) and np.array_equal(yz[:nz], g["yz"]) and np.array_equal(aFend, g["aFend"]); nchk += 1
            # PbwtCursor handed over by the device at site k, stepped on by the reference's pbwtCursorForwardsReadAD
            for k, ns in ((0, 3), (N // 2, 5), (N - 1, 1), (N, 0)):
                a_k = np.zeros(M, np.int32); d_k = np.zeros(M + 1, np.int32); y_k = np.zeros(M, np.uint8); u_k = np.zeros(M + 1, np.int32)
                a_e = np.zeros(M, np.int32); d_e = np.zeros(M + 1, np.int32); y_e = np.zeros(M, np.uint8)
                pos = (C.c_long * 4)(); c_e = C.c_int32(0)
                yzg = np.ascontiguousarray(g["yz"], np.uint8)
                vp = lambda x: x.ctypes.data_as(C.c_void_p)
                rc = r.refgpu_cursor_continue(C.c_int(M), C.c_int(N), vp(yzg), C.c_long(yzg.size), None, C.c_int(k), C.c_int(ns),
                                              vp(a_k), vp(d_k), vp(y_k), vp(u_k), pos, vp(a_e), vp(d_e), vp(y_e), C.byref(c_e))
                assert rc == 0
                assert np.array_equal(a_k, g["sweep_a"][k]) and np.array_equal(d_k, g["sweep_d"][k]) and np.array_equal(y_k, g["sweep_y"][k])
                assert pos[3] == int(g["sweep_c"][k]) and pos[2] == (1 if k < N else 0)
                assert np.array_equal(u_k, np.concatenate([[0], np.cumsum(1 - y_k.astype(np.int32))]))
                assert np.array_equal(a_e, g["sweep_a"][k + ns]) and np.array_equal(d_e, g["sweep_d"][k + ns]) and np.array_equal(y_e, g["sweep_y"][k + ns])
                assert c_e.value == int(g["sweep_c"][k + ns]); nchk += 1
        # -longWithin L through pbwtLongMatches -> matchLongWithin2 (replaced in the unity TU), with -check
        g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
        oracle.ref_long_within_file(g["yz"], 300, 400, 100, out, check=True)
        assert open(out).read() == open(os.path.join(GOLDEN, "longwithin_M300_L100.txt")).read(); nchk += 1
        # matchSequencesSweepSparse
        s = np.load(os.path.join(GOLDEN, "sparse_sweep.npz"))
        Mp, Mq, N = [int(x) for x in s["nomatch_shape"]]
        for nS in (1, 2, 3):
            assert np.array_equal(oracle.ref_match_sweep_sparse(s["nomatch_pz"], Mp, s["nomatch_qz"], Mq, N, nS), s["nomatch_s%d" % nS]); nchk += 1
        g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
        assert np.array_equal(oracle.ref_match_sweep_sparse(g["pz"], 300 - int(g["Mq"]), g["qz"], int(g["Mq"]), 400, 3), s["mosaic_M300_s3"]); nchk += 1
        # the reference's log of matchSequencesSweep on a panel with sites where no panel haplotype carries the query's allele: the
        # binding writes the "no match to query" lines (pbwtMatch.c:405-410) from the library's event list, then the averages line
        vp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.c_void_p)
        logp = os.path.join(td, "nomatch.log")
        assert r.ref_match_sweep_log_to_file(C.c_int(Mp), C.c_int(N), vp(s["nomatch_pz"]), C.c_long(len(s["nomatch_pz"])), None,
                                             C.c_int(Mq), vp(s["nomatch_qz"]), C.c_long(len(s["nomatch_qz"])), None, logp.encode()) == 0
        assert open(logp).read() == open(os.path.join(GOLDEN, "nomatch_dense.log")).read(); nchk += 1
    print("REFGPU_OK %d checks" % nchk)


if __name__ == "__main__":
    main()
