"""the host C side: `pbwt_amd/pbwt` speaks the reference's command grammar and file formats.
CPU tests: format round trips (no device work).  GPU tests: the commands that run the hot path,
byte-compared with the reference's own outputs (tests/golden) or the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, parse_pbwt


@pytest.fixture(scope="module")
def cli():
    import pbwt_amd
    return pbwt_amd.build_cli()


def run(cli, *args, check=True, cwd=None):
    r = subprocess.run([cli] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd)
    if check:
        assert r.returncode == 0, r.stderr
    return r


def test_pbwt_and_sites_roundtrip(cli, tmp_path):
    out, sites = tmp_path / "o.pbwt", tmp_path / "o.sites"
    run(cli, "-read", os.path.join(GOLDEN, "merge1.pbwt"), "-readSites", os.path.join(GOLDEN, "merge1.sites"),
        "-write", out, "-writeSites", sites)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "merge1.pbwt"), "rb").read()
    assert open(sites).read() == open(os.path.join(GOLDEN, "merge1.sites")).read()
    root = tmp_path / "all"
    run(cli, "-read", out, "-readSites", sites, "-writeAll", root)
    run(cli, "-readAll", root, "-write", tmp_path / "again.pbwt")
    assert open(tmp_path / "again.pbwt", "rb").read() == open(out, "rb").read()


def test_older_pbwt_versions_are_read(cli, tmp_path):
    """PBW2 (int byte count, no pad) and PBWT (no index arrays) headers (pbwtIO.c:182-208)"""
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "merge1.pbwt"))
    v2 = b"PBW2" + np.array([M, N], "<i4").tobytes() + aFstart.tobytes() + aFend.tobytes() + np.array([len(yz)], "<i4").tobytes() + yz.tobytes()
    open(tmp_path / "v2.pbwt", "wb").write(v2)
    run(cli, "-read", tmp_path / "v2.pbwt", "-write", tmp_path / "v3.pbwt")
    assert open(tmp_path / "v3.pbwt", "rb").read() == open(os.path.join(GOLDEN, "merge1.pbwt"), "rb").read()


def _headerless(tag, M, N, yz):
    """PBWT / GBWT files: no index arrays, a 4-byte byte count (pbwtIO.c:182-208)"""
    return tag + np.array([M, N], "<i4").tobytes() + np.array([len(yz)], "<i4").tobytes() + yz.tobytes()


def test_earliest_pbwt_versions_lack_the_end_index(cli, tmp_path):
    """`PBWT` and `GBWT` headers store neither aFstart nor aFend: the reader sets aFstart = identity and leaves aFend empty
    (pbwtIO.c:198-201), so -write refuses the panel exactly like the reference's pbwtWrite (pbwtIO.c:36)"""
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "merge1.pbwt"))
    for tag in (b"PBWT", b"GBWT"):
        f = tmp_path / (tag.decode() + ".pbwt")
        f.write_bytes(_headerless(tag, M, N, yz))
        r = run(cli, "-read", f, "-write", tmp_path / "out.pbwt", check=False)
        assert r.returncode != 0 and "pbwtWrite called without start and end indexes" in r.stderr
        assert ("read pbwt %s file with %d bytes: M, N are %d, %d" % (tag.decode(), len(yz), M, N)) in r.stderr


@pytest.mark.gpu
def test_earliest_pbwt_versions_sweep_like_v3(cli, tmp_path):
    """the same panels through the device: -haps and -maxWithin of a `PBWT` / `GBWT` file equal those of the PBW3 file (merge1's start
    order is the identity the old formats imply) — the reference's own test vector test/merge.1.out among them"""
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "merge1.pbwt"))
    assert np.array_equal(aFstart, np.arange(M))
    for tag in (b"PBWT", b"GBWT"):
        f = tmp_path / (tag.decode() + ".pbwt")
        f.write_bytes(_headerless(tag, M, N, yz))
        run(cli, "-read", f, "-haps", tmp_path / "h.txt")
        assert open(tmp_path / "h.txt").read() == open(os.path.join(GOLDEN, "merge1.haps")).read()
        r = run(cli, "-read", f, "-maxWithin")
        assert r.stdout == open(os.path.join(GOLDEN, "merge1.maxwithin.txt")).read()


def _macs_text(bits, M, N):
    """a MaCS output file (pbwtIO.c:426-458 grammar) for the panel `bits` (N bit columns)"""
    hap = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :M]
    rows = (hap + ord("0")).astype(np.uint8)
    out = ["COMMAND:\tmacs %d 1e6 -t 0.001 -r 0.001" % M, "SEED:\t1"]
    body = [b"SITE:\t%d\t%.10f\t0.1\t" % (k, (k + 0.5) / N) + rows[k].tobytes() for k in range(N)]
    return ("\n".join(out) + "\n").encode() + b"\n".join(body) + b"\n"


@pytest.mark.gpu
def test_config0_full_size_readMacs_write_maxWithin(cli, orc, tmp_path):
    """BASELINE configs[0] at its own size: a 2 000-haplotype x 20 000-site MaCS text file through `-readMacs -write -maxWithin`
    (pbwtIO.c:460-492 + pbwtMatch.c:115-142 on the device).  The .pbwt bytes equal the oracle's (and, where its library travelled
    with the repo, the reference's own pbwtReadMacs + pbwtWrite); the 3.9 M MATCH lines equal the oracle's report stream."""
    import hashlib
    M, N = 2000, 20000
    bits = orc.synth_bitcols(M, N, seed=20, kind=0)
    macs = tmp_path / "c0.macs"
    macs.write_bytes(_macs_text(bits, M, N))
    out = tmp_path / "c0.pbwt"
    r = subprocess.run([cli, "-readMacs", str(macs), "-write", str(out), "-maxWithin"], capture_output=True)
    assert r.returncode == 0, r.stderr[-2000:]
    o = orc.build_bitcols(bits, M, with_d=False)
    _M, _N, a0, a1, yz = parse_pbwt(out)
    assert (_M, _N) == (M, N) and np.array_equal(yz, o["yz"]) and np.array_equal(a1, o["aFend"]) and np.array_equal(a0, np.arange(M))
    if orc.ref() is not None:                                  # the reference itself on the same text
        import ctypes as C
        rp, rs = tmp_path / "ref.pbwt", tmp_path / "ref.sites"
        assert orc.ref().ref_macs_to_pbwt(str(macs).encode(), str(rp).encode(), str(rs).encode()) == 0
        assert open(rp, "rb").read() == open(out, "rb").read()
    recs = orc.max_within(o["yz"], M, N)
    keep = recs[recs["start"] != recs["end"]]                   # reportMatch drops zero-length matches (pbwtMatch.c:48)
    want = hashlib.sha256()
    for lo in range(0, len(keep), 200000):
        part = keep[lo: lo + 200000]
        want.update("".join("MATCH\t%d\t%d\t%d\t%d\t%d\n" % (m[0], m[1], m[2], m[3], m[3] - m[2]) for m in part.tolist()).encode())
    assert r.stdout.count(b"\n") == len(keep) and len(keep) > 1000000
    assert hashlib.sha256(r.stdout).hexdigest() == want.hexdigest()


def test_errors_die_like_the_reference(cli, tmp_path):
    r = run(cli, "-write", tmp_path / "x", check=False)
    assert r.returncode != 0 and r.stderr.startswith("FATAL ERROR: ")
    bad = tmp_path / "bad.pbwt"
    bad.write_bytes(b"NOPE" + b"\0" * 16)
    r = run(cli, "-read", bad, check=False)
    assert r.returncode != 0 and "failed to recognise file type" in r.stderr
    r = run(cli, "-read", os.path.join(GOLDEN, "merge1.pbwt"), "-readSites", os.path.join(GOLDEN, "macs_small.sites"), check=False)
    assert r.returncode != 0 and "sites file contains" in r.stderr


@pytest.mark.gpu
def test_readMacs_write_matches_reference_bytes(cli, tmp_path):
    """BASELINE configs[0] plumbing: -readMacs -write -writeSites reproduces the reference's files"""
    out, sites = tmp_path / "m.pbwt", tmp_path / "m.sites"
    run(cli, "-readMacs", os.path.join(GOLDEN, "macs_small.macs"), "-write", out, "-writeSites", sites)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "macs_small.pbwt"), "rb").read()
    assert open(sites).read() == open(os.path.join(GOLDEN, "macs_small.sites")).read()


@pytest.mark.gpu
def test_checkpoint_files_match_reference_bytes(cli, tmp_path):
    """-checkpoint n (pbwtIO.c:27,158-168,481): check_A / check_B .pbwt + .sites dropped into the working directory
    every n sites, alternating; bytes as the reference writes them, and the final panel unchanged"""
    out = tmp_path / "m.pbwt"
    run(cli, "-checkpoint", 50, "-readMacs", os.path.join(GOLDEN, "macs_small.macs"), "-write", out, cwd=tmp_path)
    for ab in "AB":
        assert open(tmp_path / ("check_%s.pbwt" % ab), "rb").read() == open(os.path.join(GOLDEN, "macs_small.check_%s.pbwt" % ab), "rb").read()
        assert open(tmp_path / ("check_%s.sites" % ab)).read() == open(os.path.join(GOLDEN, "macs_small.check_%s.sites" % ab)).read()
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "macs_small.pbwt"), "rb").read()
    # a checkpoint is a complete panel of the first n sites
    r = run(cli, "-read", tmp_path / "check_A.pbwt", "-readSites", tmp_path / "check_A.sites", "-haps", tmp_path / "a.haps")
    assert len(open(tmp_path / "a.haps").read().splitlines()) == 50


@pytest.mark.gpu
def test_maxWithin_and_haps_text(cli, tmp_path):
    r = run(cli, "-check", "-read", os.path.join(GOLDEN, "merge1.pbwt"), "-maxWithin")
    assert r.stdout == open(os.path.join(GOLDEN, "merge1.maxwithin.txt")).read()
    run(cli, "-read", os.path.join(GOLDEN, "merge1.pbwt"), "-haps", tmp_path / "h")
    assert open(tmp_path / "h").read() == open(os.path.join(GOLDEN, "merge1.haps")).read()


@pytest.mark.gpu
def test_stats_histogram_text(cli, tmp_path):
    g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
    M, N = int(g["M"]), int(g["N"])
    f = tmp_path / "p.pbwt"
    f.write_bytes(b"PBW3" + np.array([M, N], "<i4").tobytes() + np.arange(M, dtype="<i4").tobytes() + g["aFend"].astype("<i4").tobytes()
                  + np.array([len(g["yz"])], "<i8").tobytes() + b"    " + g["yz"].tobytes())
    r = run(cli, "-stats", "-read", f, "-maxWithin")
    assert r.stdout == g["hist_txt"].tobytes().decode()
    assert "Average" in r.stderr


@pytest.mark.gpu
def test_matchDynamic_and_siteInfo_and_subsample(cli, orc, tmp_path):
    g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
    M, N, Mq = int(g["M"]), int(g["N"]), int(g["Mq"])

    def write(path, m, yz, aFend=None):
        path.write_bytes(b"PBW3" + np.array([m, N], "<i4").tobytes() + np.arange(m, dtype="<i4").tobytes()
                         + (np.zeros(m, "<i4") if aFend is None else aFend.astype("<i4")).tobytes()
                         + np.array([len(yz)], "<i8").tobytes() + b"    " + yz.tobytes())
    write(tmp_path / "p.pbwt", M - Mq, g["pz"]); write(tmp_path / "q.pbwt", Mq, g["qz"]); write(tmp_path / "full.pbwt", M, g["yz"], g["aFend"])
    r = run(cli, "-check", "-read", tmp_path / "p.pbwt", "-matchDynamic", tmp_path / "q.pbwt")
    want = "".join("MATCH\t%d\t%d\t%d\t%d\t%d\n" % (a, b, s, e, e - s) for (a, b, s, e) in g["qrecs"].tolist() if s != e)
    assert r.stdout == want
    assert "Average number of best matches including alternates" in r.stderr
    # -subsample start n == the reference's -subsample: the panel rebuilt from those haplotypes
    run(cli, "-read", tmp_path / "full.pbwt", "-subsample", M - Mq, Mq, "-write", tmp_path / "sub.pbwt")
    _, _, _, aFend, yz = parse_pbwt(str(tmp_path / "sub.pbwt"))
    hap = orc.unpack_bitcols(g["bits"], M)
    o = orc.build_bitcols(orc.pack_bitcols(hap[:, M - Mq:]), Mq, with_d=False)
    assert np.array_equal(yz, o["yz"]) and np.array_equal(aFend, o["aFend"])
    # -siteInfo f kmin kmax: "y (k - d)" pairs of the selected sites (pbwtMain.c:82-100)
    run(cli, "-read", tmp_path / "full.pbwt", "-siteInfo", tmp_path / "si.txt", 20, 60)
    sw = orc.sweep_AD(g["yz"], M, N, dump_sites=range(N))
    lines = []
    for k in range(N):
        f1 = int(sw["y_dump"][k].sum())
        if 20 <= f1 < 60:
            lines.append("".join("%d %d " % (sw["y_dump"][k][j], k - sw["d_dump"][k][j]) for j in range(M)) + "\n")
    assert len(lines) > 3 and open(tmp_path / "si.txt").read() == "".join(lines)


@pytest.mark.gpu
def test_longWithin_text(cli, tmp_path):
    g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
    M, N = int(g["M"]), int(g["N"])
    f = tmp_path / "p.pbwt"
    f.write_bytes(b"PBW3" + np.array([M, N], "<i4").tobytes() + np.arange(M, dtype="<i4").tobytes() + g["aFend"].astype("<i4").tobytes()
                  + np.array([len(g["yz"])], "<i8").tobytes() + b"    " + g["yz"].tobytes())
    r = run(cli, "-check", "-read", f, "-longWithin", 100)
    assert r.stdout == open(os.path.join(GOLDEN, "longwithin_M300_L100.txt")).read()


@pytest.mark.gpu
def test_buildReverse_matches_reference_bytes(cli, tmp_path):
    """-read x -buildReverse -writeReverse y == the reference's reverse PBWT (zz, aRstart, aRend)"""
    run(cli, "-read", os.path.join(GOLDEN, "macs_small.pbwt"), "-buildReverse", "-writeReverse", tmp_path / "rev.pbwt")
    assert open(tmp_path / "rev.pbwt", "rb").read() == open(os.path.join(GOLDEN, "macs_small.reverse.pbwt"), "rb").read()
    # and it reads back as the reverse of the same panel
    run(cli, "-read", os.path.join(GOLDEN, "macs_small.pbwt"), "-readReverse", tmp_path / "rev.pbwt", "-writeReverse", tmp_path / "rev2.pbwt")
    assert open(tmp_path / "rev2.pbwt", "rb").read() == open(tmp_path / "rev.pbwt", "rb").read()


def write_pbwt(path, M, N, yz):
    ident = np.arange(M, dtype="<i4").tobytes()
    open(path, "wb").write(b"PBW3" + np.array([M, N], "<i4").tobytes() + ident + ident + np.array([len(yz)], "<i8").tobytes() + b"    " + np.asarray(yz, np.uint8).tobytes())


@pytest.mark.gpu
def test_log_lines_match_the_reference(cli, tmp_path):
    """what log scrapers see: the "no match to query" events of -matchDynamic, one line each in the reference's order
    (golden written by the reference: tests/golden/nomatch_dense.log), the averages line, and the timing line after
    every command in the format of utils.c:173-198"""
    import re
    s = np.load(os.path.join(GOLDEN, "sparse_sweep.npz"))
    Mp, Mq, N = [int(x) for x in s["nomatch_shape"]]
    write_pbwt(tmp_path / "p.pbwt", Mp, N, s["nomatch_pz"])
    write_pbwt(tmp_path / "q.pbwt", Mq, N, s["nomatch_qz"])
    r = run(cli, "-log", tmp_path / "log.txt", "-read", tmp_path / "p.pbwt", "-matchDynamic", tmp_path / "q.pbwt")
    log = open(tmp_path / "log.txt").read().splitlines()
    want = open(os.path.join(GOLDEN, "nomatch_dense.log")).read().splitlines()
    assert [ln for ln in log if ln.startswith("no match") or ln.startswith("Average number")] == want
    assert r.stdout == "".join("MATCH\t%d\t%d\t%d\t%d\t%d\n" % (m["ai"], m["bi"], m["start"], m["end"], m["end"] - m["start"])
                               for m in s["nomatch_dense"] if m["start"] != m["end"])
    # the sites-file readers log one line per file (pbwtIO.c:263): -readSites and the list of -selectSites alike
    P, S, lst = (os.path.join(GOLDEN, f) for f in ("macs_small.pbwt", "macs_small.sites", "macs_small.select.sites"))
    run(cli, "-log", tmp_path / "log2.txt", "-read", P, "-readSites", S, "-selectSites", lst, "-write", tmp_path / "sel.pbwt")
    got = [ln for ln in open(tmp_path / "log2.txt").read().splitlines() if ln.startswith("read ") and " sites on chromosome " in ln]
    assert got == open(os.path.join(GOLDEN, "macs_small.select.log")).read().splitlines()
    timing = [ln for ln in log if ln.startswith("user\t")]
    assert len(timing) == 3 and all(re.fullmatch(r"user\t\d+\.\d{6}\tsystem\t\d+\.\d{6}\tmax_RSS\t-?\d+\tMemory\t\d+", ln) for ln in timing)


@pytest.mark.gpu
def test_stats_after_longWithin_and_writeAll_reverse(cli, tmp_path):
    """-stats -longWithin L: the MATCH lines, then the stats block over an empty histogram (pbwtMatch.c:166-178: no bins,
    0.0 matches per sample, 0/0 average length); -buildReverse -writeAll / -readAll carry root.reverse (pbwtIO.c:143,419)"""
    f = os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz")
    g = np.load(f)
    write_pbwt(tmp_path / "m.pbwt", 300, 400, g["yz"])
    r = run(cli, "-stats", "-read", tmp_path / "m.pbwt", "-longWithin", 100)
    assert r.stdout == open(os.path.join(GOLDEN, "longwithin_M300_L100.txt")).read()
    assert "Average 0.0 matches per sample" in r.stderr and "Average length -nan" in r.stderr
    root = tmp_path / "all"
    run(cli, "-read", os.path.join(GOLDEN, "macs_small.pbwt"), "-buildReverse", "-writeAll", root)
    assert open(str(root) + ".reverse", "rb").read() == open(os.path.join(GOLDEN, "macs_small.reverse.pbwt"), "rb").read()
    run(cli, "-readAll", root, "-writeReverse", tmp_path / "again.reverse")
    assert open(tmp_path / "again.reverse", "rb").read() == open(os.path.join(GOLDEN, "macs_small.reverse.pbwt"), "rb").read()


@pytest.mark.gpu
def test_panel_transforms_match_reference_bytes(cli, gpu_lib, tmp_path):
    """-subrange / -selectSites / -removeSites (incl. its stop-with-the-list quirk) and pbwtSubSample with an arbitrary
    selection: decode, regather and rebuild on the device (pbwtamd_regather); .pbwt and .sites bytes are the reference's"""
    P, S = os.path.join(GOLDEN, "macs_small.pbwt"), os.path.join(GOLDEN, "macs_small.sites")
    lst = os.path.join(GOLDEN, "macs_small.select.sites")
    for tag, args in (("subrange", ["-subrange", 17, 93]), ("selected", ["-selectSites", lst]), ("removed", ["-removeSites", lst])):
        out, outs = tmp_path / (tag + ".pbwt"), tmp_path / (tag + ".sites")
        run(cli, "-read", P, "-readSites", S, *args, "-write", out, "-writeSites", outs)
        assert open(out, "rb").read() == open(os.path.join(GOLDEN, "macs_small.%s.pbwt" % tag), "rb").read(), tag
        assert open(outs).read() == open(os.path.join(GOLDEN, "macs_small.%s.out.sites" % tag)).read(), tag
    # the general selection goes through the ABI (the CLI only has the interval form)
    M, N, aFstart, aFend, yz = parse_pbwt(P)
    sel = np.load(os.path.join(GOLDEN, "macs_small.subsample10.select.npy"))
    eng = gpu_lib.Engine(M, batch_sites=32)
    r = eng.regather(yz, N, hap_select=sel, aFstart=aFstart)
    M2, N2, a0, a1, yz2 = parse_pbwt(os.path.join(GOLDEN, "macs_small.subsample10.pbwt"))
    assert (M2, N2) == (len(sel), N) and np.array_equal(r["yz"], yz2) and np.array_equal(r["aFend"], a1)
    # the interval form of the CLI against the same ABI call
    run(cli, "-read", P, "-subsample", 20, 25, "-write", tmp_path / "ss.pbwt")
    M3, N3, b0, b1, yz3 = parse_pbwt(tmp_path / "ss.pbwt")
    r3 = eng.regather(yz, N, hap_select=np.arange(20, 45), aFstart=aFstart)
    assert M3 == 25 and np.array_equal(r3["yz"], yz3) and np.array_equal(r3["aFend"], b1)
    # reverse panel through the same entry point: site order N-1..0, started from the forward panel's final order
    rev = eng.regather(yz, N, site_order=np.arange(N - 1, -1, -1), aFstart=aFstart, aStart_out=aFend, want_fwd_end=True)
    Mr, Nr, r0, r1, zz = parse_pbwt(os.path.join(GOLDEN, "macs_small.reverse.pbwt"))
    assert np.array_equal(rev["aFend_fwd"], aFend) and np.array_equal(rev["yz"], zz) and np.array_equal(rev["aFend"], r1)
