"""Generate the golden fixtures under tests/golden/ with the REAL reference (oracle/_ref, compiled
in place from /root/reference).  Run in the build container only:  python tests/golden/make_golden.py

Fixtures are data only (inputs + the reference's outputs):
  merge1.*            the reference's own tiny test panel (test/merge.1.tab, 8 haplotypes x 10 sites):
                      .pbwt bytes written by pbwtWrite, -haps text, per-site a/d/y dumps, MATCH records
  mosaic_M*_N*.npz    synthetic panels (generator recipe in oracle/pbwt_oracle.c): bit columns, packed
                      PBWT, aFend, per-site a/d, matchMaximalWithin records, -stats histogram text,
                      matchSequencesSweep records for a held-out query split
  macs_small.*        a MaCS-format text panel and the .pbwt/.sites the reference builds from it
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import ctypes as C  # noqa: E402

ref = oracle.ref()
assert ref is not None, "oracle/_ref not built (needs /root/reference)"


def parse_pbwt(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"PBW3"
    M, N = np.frombuffer(raw, "<i4", 2, 4)
    off = 12
    aFstart = np.frombuffer(raw, "<i4", M, off); off += 4 * M
    aFend = np.frombuffer(raw, "<i4", M, off); off += 4 * M
    nz = int(np.frombuffer(raw, "<i8", 1, off)[0]); off += 8 + 4
    yz = np.frombuffer(raw, np.uint8, nz, off)
    return int(M), int(N), aFstart.copy(), aFend.copy(), yz.copy()


def mosaic(M, N, seed, kind, qsplit):
    bits = oracle.synth_bitcols(M, N, seed=seed, kind=kind)
    hap = oracle.unpack_bitcols(bits, M)
    rb = oracle.ref_build_bitcols(bits, M, with_d=True)
    rbA = oracle.ref_build_bitcols(bits, M, with_d=False)
    assert np.array_equal(rb["yz"], rbA["yz"]) and np.array_equal(rb["aFend"], rbA["aFend"])
    yz = rb["yz"]
    sw = oracle.ref_sweep_dump(yz, M, N)
    recs = oracle.ref_max_within(yz, M, N)
    hist_path = os.path.join(HERE, "_tmp_hist.txt")
    oracle.ref_max_within_file(yz, M, N, hist_path, hist=True, check=True)
    hist_txt = open(hist_path).read(); os.remove(hist_path)
    Mq = qsplit; Mp = M - Mq
    pb = oracle.pack_bitcols(hap[:, :Mp]); qb = oracle.pack_bitcols(hap[:, Mp:])
    pz = oracle.ref_build_bitcols(pb, Mp, with_d=False)["yz"]
    qz = oracle.ref_build_bitcols(qb, Mq, with_d=False)["yz"]
    qrecs = oracle.ref_match_sweep(pz, Mp, qz, Mq, N)
    if M == 300:      # -longWithin 100 text exactly as the reference CLI prints it (with -check)
        oracle.ref_long_within_file(yz, M, N, 100, os.path.join(HERE, "longwithin_M300_L100.txt"), check=True)
    name = os.path.join(HERE, "mosaic_M%d_N%d_k%d.npz" % (M, N, kind))
    np.savez_compressed(name, M=M, N=N, seed=seed, kind=kind, bits=bits, yz=yz, aFend=rb["aFend"],
                        build_a=rb["a_all"].astype(np.int32), build_d=rb["d_all"].astype(np.int32),
                        sweep_a=sw["a_all"], sweep_d=sw["d_all"], sweep_y=sw["y_all"], sweep_c=sw["c_all"],
                        within=recs, hist_txt=np.frombuffer(hist_txt.encode(), np.uint8),
                        Mq=Mq, pz=pz, qz=qz, qrecs=qrecs)
    print("wrote", name, "within", len(recs), "qrecs", len(qrecs))


def sparse_sweep():
    """matchSequencesSweepSparse (pbwtMatch.c:501-602) records of the reference: the query splits of two mosaic
    goldens at nSparse = 2, 3, 4 and a crafted panel with a site where no panel haplotype carries the query's allele
    (the 'no match to query' branch, dense and sparse)"""
    out = {}
    for name in ("mosaic_M70_N150_k1.npz", "mosaic_M300_N400_k0.npz"):
        g = np.load(os.path.join(HERE, name))
        M, N, Mq = int(g["M"]), int(g["N"]), int(g["Mq"])
        for nS in (2, 3, 4):
            out["%s_s%d" % (name.split("_N")[0], nS)] = oracle.ref_match_sweep_sparse(g["pz"], M - Mq, g["qz"], Mq, N, nS)
    rng = np.random.default_rng(5)
    Mp, Mq, N = 20, 4, 31
    hap = (rng.random((N, Mp + Mq)) < 0.5).astype(np.uint8)
    for k in (0, 7, 8, 20):
        hap[k, :Mp] = 0; hap[k, Mp:] = [1, 0, 1, 1]
    hap[13, :Mp] = 1; hap[13, Mp:] = [0, 0, 1, 0]
    pz = oracle.ref_build_bitcols(oracle.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qz = oracle.ref_build_bitcols(oracle.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    out["nomatch_pz"] = pz; out["nomatch_qz"] = qz
    out["nomatch_shape"] = np.array([Mp, Mq, N])
    for nS in (1, 2, 3):
        out["nomatch_s%d" % nS] = oracle.ref_match_sweep_sparse(pz, Mp, qz, Mq, N, nS)
    out["nomatch_dense"] = oracle.ref_match_sweep(pz, Mp, qz, Mq, N)
    # the reference's log for that sweep: one "no match to query ..." line per event, then the averages line
    vp = lambda x: np.ascontiguousarray(x).ctypes.data_as(C.c_void_p)
    logp = os.path.join(HERE, "nomatch_dense.log")
    assert ref.ref_match_sweep_log_to_file(C.c_int(Mp), C.c_int(N), vp(pz), C.c_long(len(pz)), None,
                                           C.c_int(Mq), vp(qz), C.c_long(len(qz)), None, logp.encode()) == 0
    np.savez_compressed(os.path.join(HERE, "sparse_sweep.npz"), **out)
    print("wrote sparse_sweep.npz", {k: len(v) for k, v in out.items()})


def merge1():
    tab = "/root/reference/test/merge.1.tab"
    out = os.path.join(HERE, "merge1.pbwt")
    assert ref.ref_vcfq_to_pbwt(tab.encode(), out.encode(), os.path.join(HERE, "merge1.sites").encode()) == 0
    assert ref.ref_pbwt_to_haps(out.encode(), os.path.join(HERE, "merge1.haps").encode()) == 0
    M, N, aFstart, aFend, yz = parse_pbwt(out)
    sw = oracle.ref_sweep_dump(yz, M, N, aFstart)
    recs = oracle.ref_max_within(yz, M, N, aFstart)
    txt = os.path.join(HERE, "merge1.maxwithin.txt")
    oracle.ref_max_within_file(yz, M, N, txt, aFstart=aFstart, check=True)
    np.savez_compressed(os.path.join(HERE, "merge1.npz"), M=M, N=N, yz=yz, aFstart=aFstart, aFend=aFend,
                        sweep_a=sw["a_all"], sweep_d=sw["d_all"], sweep_y=sw["y_all"], sweep_c=sw["c_all"], within=recs)
    # the committed .haps must equal the reference's own golden for this panel
    assert open(os.path.join(HERE, "merge1.haps")).read() == open("/root/reference/test/merge.1.out").read()
    print("wrote merge1.*", M, N, len(recs))


def macs_small():
    M, N = 60, 120
    bits = oracle.synth_bitcols(M, N, seed=99, kind=0)
    hap = oracle.unpack_bitcols(bits, M)
    path = os.path.join(HERE, "macs_small.macs")
    with open(path, "w") as f:
        f.write("COMMAND:\tmacs %d 1e6 -t 0.001 -r 0.001\nSEED:\t1\n" % M)
        for k in range(N):
            f.write("SITE:\t%d\t%.8f\t0.1\t%s\n" % (k, (k + 0.5) / N, "".join(map(str, hap[k]))))
    assert ref.ref_macs_to_pbwt(path.encode(), os.path.join(HERE, "macs_small.pbwt").encode(),
                                os.path.join(HERE, "macs_small.sites").encode()) == 0
    assert ref.ref_build_reverse(os.path.join(HERE, "macs_small.pbwt").encode(), os.path.join(HERE, "macs_small.reverse.pbwt").encode()) == 0
    # -checkpoint 50: the reference drops check_A (50 sites) and check_B (100 sites) into the working directory
    import tempfile, shutil
    with tempfile.TemporaryDirectory() as td:
        assert ref.ref_macs_checkpoint(path.encode(), 50, td.encode()) == 0
        for ab in "AB":
            for ext in ("pbwt", "sites"):
                shutil.copy(os.path.join(td, "check_%s.%s" % (ab, ext)), os.path.join(HERE, "macs_small.check_%s.%s" % (ab, ext)))
    print("wrote macs_small.*")


def transforms():
    """panel transforms of the reference on macs_small (.pbwt + .sites): -subrange 17 93, -selectSites / -removeSites with a
    list holding every third site plus a position the panel lacks, pbwtSubSample with a 10-haplotype selection"""
    P = os.path.join(HERE, "macs_small.pbwt").encode(); S = os.path.join(HERE, "macs_small.sites").encode()
    sites = open(os.path.join(HERE, "macs_small.sites")).read().splitlines()
    lst = [sites[i] for i in range(0, len(sites), 3)]
    chrom = sites[0].split("\t")[0]
    lst.insert(5, "%s\t%d\t%s" % (chrom, int(sites[14].split("\t")[1]) + 1, sites[14].split("\t", 2)[2]))
    lst = sorted(lst, key=lambda l: int(l.split("\t")[1]))
    open(os.path.join(HERE, "macs_small.select.sites"), "w").write("\n".join(lst) + "\n")
    ref.ref_transform.restype = C.c_int

    def run(op, i0=0, i1=0, lf=None, sel=None, tag=""):
        sel_a = np.asarray(sel if sel is not None else [0], np.int32)
        rc = ref.ref_transform(P, S, C.c_int(op), C.c_int(i0), C.c_int(i1), lf.encode() if lf else None, sel_a.ctypes.data_as(C.c_void_p),
                               C.c_int(len(sel_a)), os.path.join(HERE, "macs_small.%s.pbwt" % tag).encode(),
                               os.path.join(HERE, "macs_small.%s.out.sites" % tag).encode())
        assert rc == 0, (tag, rc)
    run(0, 17, 93, tag="subrange")
    run(1, lf=os.path.join(HERE, "macs_small.select.sites"), tag="selected")
    run(2, lf=os.path.join(HERE, "macs_small.select.sites"), tag="removed")
    sel = [59, 3, 4, 40, 7, 22, 0, 31, 58, 12]
    run(3, sel=sel, tag="subsample10")
    np.save(os.path.join(HERE, "macs_small.subsample10.select.npy"), np.asarray(sel, np.int32))
    # what the reference logs while it reads the two sites files of `-readSites S -selectSites L`
    assert ref.ref_read_sites_log(P, S, os.path.join(HERE, "macs_small.select.sites").encode(), os.path.join(HERE, "macs_small.select.log").encode()) == 0
    print("wrote macs_small.{subrange,selected,removed,subsample10}.*")


if __name__ == "__main__":
    macs_small()      # first: the reference's global variation dict must still be empty (fresh-process behaviour)
    merge1()
    mosaic(8, 10, 3, 1, 2)
    mosaic(70, 150, 11, 1, 10)       # iid, M not a multiple of 64
    mosaic(300, 400, 5, 0, 40)       # founder mosaic
    mosaic(1100, 260, 6, 0, 100)     # spans two 1024-position tiles
    sparse_sweep()                   # needs the mosaic goldens above
    transforms()                     # needs macs_small.*
