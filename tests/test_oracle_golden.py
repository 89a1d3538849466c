"""CPU: the oracle restatement against the committed golden vectors (made by the real reference,
tests/golden/make_golden.py) and, where oracle/_ref is present, against the reference directly."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_panels, parse_pbwt


def test_known_answer_vector_survey(orc):
    """8 haplotypes x 10 sites = the reference's test/merge.1.tab"""
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "merge1.pbwt"))
    assert (M, N) == (8, 10)
    assert yz.tobytes().hex(" ").startswith("81 04 81 02 81 03 81 03")
    sw = orc.sweep_AD(yz, M, N, aFstart, dump_sites=range(N + 1))
    assert sw["a_dump"][1].tolist() == [1, 2, 3, 4, 6, 7, 0, 5]
    assert sw["d_dump"][9].tolist() == [10, 4, 5, 6, 8, 3, 9, 7, 10]
    assert sw["a_dump"][10].tolist() == aFend.tolist() == [3, 4, 0, 7, 1, 6, 2, 5]
    g = np.load(os.path.join(GOLDEN, "merge1.npz"))
    assert np.array_equal(sw["a_dump"], g["sweep_a"]) and np.array_equal(sw["d_dump"], g["sweep_d"])
    assert np.array_equal(sw["y_dump"], g["sweep_y"]) and np.array_equal(sw["c_dump"], g["sweep_c"])
    recs = orc.max_within(yz, M, N, aFstart)
    assert np.array_equal(recs, g["within"])
    # CLI text of -maxWithin (reportMatch drops zero-length matches, pbwtMatch.c:48)
    lines = ["MATCH\t%d\t%d\t%d\t%d\t%d\n" % (r["ai"], r["bi"], r["start"], r["end"], r["end"] - r["start"])
             for r in recs if r["start"] != r["end"]]
    assert "".join(lines) == open(os.path.join(GOLDEN, "merge1.maxwithin.txt")).read()
    assert len(lines) == 24
    # -haps round trip = the reference's own golden merge.1.out
    hap = orc.haplotypes(yz, M, N, aFstart)
    txt = "".join("".join(map(str, row)) + "\n" for row in hap)
    assert txt == open(os.path.join(GOLDEN, "merge1.haps")).read()


def test_macs_build_matches_reference_pbwt(orc):
    """-readMacs ... -write: build loop over the MaCS panel reproduces the reference's .pbwt bytes"""
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "macs_small.pbwt"))
    rows = [ln.split("\t")[4].strip() for ln in open(os.path.join(GOLDEN, "macs_small.macs")) if ln.startswith("SITE:")]
    hap = np.array([[int(c) for c in r] for r in rows], dtype=np.uint8)
    assert hap.shape == (N, M)
    out = orc.build_bitcols(orc.pack_bitcols(hap), M, with_d=False)
    assert np.array_equal(out["yz"], yz) and np.array_equal(out["aFend"], aFend)
    assert np.array_equal(aFstart, np.arange(M))


@pytest.mark.parametrize("path", golden_panels(), ids=os.path.basename)
def test_oracle_vs_golden_panel(orc, path):
    g = np.load(path)
    M, N = int(g["M"]), int(g["N"])
    bits = g["bits"]
    # the generator itself
    assert np.array_equal(orc.synth_bitcols(M, N, seed=int(g["seed"]), kind=int(g["kind"])), bits)
    b = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    assert np.array_equal(b["yz"], g["yz"]) and np.array_equal(b["aFend"], g["aFend"])
    assert np.array_equal(b["a_dump"], g["build_a"]) and np.array_equal(b["d_dump"], g["build_d"])
    for k in (0, N // 2, N):
        assert b["csum_a"][k] == orc.checksum_i32(g["build_a"][k])
        assert b["csum_d"][k] == orc.checksum_i32(g["build_d"][k])
    bA = orc.build_bitcols(bits, M, with_d=False)
    assert np.array_equal(bA["yz"], g["yz"]) and np.array_equal(bA["aFend"], g["aFend"])
    sw = orc.sweep_AD(g["yz"], M, N, dump_sites=range(N + 1))
    assert np.array_equal(sw["a_dump"], g["sweep_a"]) and np.array_equal(sw["d_dump"], g["sweep_d"])
    assert np.array_equal(sw["y_dump"], g["sweep_y"]) and np.array_equal(sw["c_dump"], g["sweep_c"])
    assert np.array_equal(orc.max_within(g["yz"], M, N), g["within"])
    hist = orc.max_within_hist(g["yz"], M, N)
    txt = "".join("%d\t%d\n" % (i, c) for i, c in enumerate(hist) if c)
    assert txt == g["hist_txt"].tobytes().decode()
    Mq = int(g["Mq"])
    recs, nomatch, tot = orc.match_sweep(g["pz"], M - Mq, g["qz"], Mq, N)
    assert np.array_equal(recs, g["qrecs"])
    assert tot[0] == len(recs)


def test_pack3_edge_cases(orc):
    rng = np.random.default_rng(1)
    for M in (1, 2, 63, 64, 65, 2047, 2048, 2049, 63487, 63488, 63489, 70000, 130000):
        for y in (np.zeros(M, np.uint8), np.ones(M, np.uint8), (rng.random(M) < 0.01).astype(np.uint8),
                  (rng.random(M) < 0.5).astype(np.uint8)):
            z = orc.pack3(y)
            back, used, n0 = orc.unpack3(z, M)
            assert used == len(z) and np.array_equal(back, y) and n0 == int((y == 0).sum())
    # byte-level spot checks of the three run classes (pbwtCore.c:216-225)
    assert orc.pack3(np.zeros(5, np.uint8)).tolist() == [5]
    assert orc.pack3(np.ones(64, np.uint8)).tolist() == [0x80 | 0x40 | 1]
    assert orc.pack3(np.ones(2048 + 64 + 3, np.uint8)).tolist() == [0x80 | 0x60 | 1, 0x80 | 0x40 | 1, 0x80 | 3]
    assert orc.pack3(np.zeros(63488 + 1, np.uint8)).tolist() == [0x7f, 1]


def test_oracle_vs_real_reference_random(orc):
    """only where oracle/_ref exists (build container, or the prebuilt .so shipped to the GPU box)"""
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    for (M, N, kind, seed) in [(2, 7, 1, 1), (5, 33, 1, 2), (129, 77, 1, 3), (513, 90, 0, 4)]:
        bits = orc.synth_bitcols(M, N, seed=seed, kind=kind)
        o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
        r = orc.ref_build_bitcols(bits, M, with_d=True)
        assert np.array_equal(o["yz"], r["yz"]) and np.array_equal(o["a_dump"], r["a_all"]) and np.array_equal(o["d_dump"], r["d_all"])
        assert np.array_equal(orc.max_within(o["yz"], M, N), orc.ref_max_within(o["yz"], M, N))


def test_long_within_oracle_vs_reference_text(orc):
    """-longWithin 100 on the 300-haplotype golden panel: the oracle's records print to the reference's text"""
    g = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
    recs = orc.long_within(g["yz"], int(g["M"]), int(g["N"]), 100)
    txt = "".join("MATCH\t%d\t%d\t%d\t%d\t%d\n" % (r["ai"], r["bi"], r["start"], r["end"], r["end"] - r["start"]) for r in recs if r["start"] != r["end"])
    assert txt == open(os.path.join(GOLDEN, "longwithin_M300_L100.txt")).read()


def test_sparse_sweep_matches_reference_goldens(orc):
    """matchSequencesSweepSparse (pbwtMatch.c:501-602): the oracle's restatement against records captured from the
    reference (tests/golden/sparse_sweep.npz, made by make_golden.py::sparse_sweep), incl. the 'no match to query'
    branch; nSparse = 1 degenerates to the dense sweep"""
    g = np.load(os.path.join(GOLDEN, "sparse_sweep.npz"))
    for name in ("mosaic_M70_N150_k1.npz", "mosaic_M300_N400_k0.npz"):
        m = np.load(os.path.join(GOLDEN, name))
        M, N, Mq = int(m["M"]), int(m["N"]), int(m["Mq"])
        for nS in (2, 3, 4):
            recs, nomatch, tot = orc.match_sweep_sparse(m["pz"], M - Mq, m["qz"], Mq, N, nS)
            want = g["%s_s%d" % (name.split("_N")[0], nS)]
            assert np.array_equal(recs, want.view(recs.dtype).reshape(-1)) and tot[0] == len(recs)
    Mp, Mq, N = (int(v) for v in g["nomatch_shape"])
    for nS in (1, 2, 3):
        recs, nomatch, tot = orc.match_sweep_sparse(g["nomatch_pz"], Mp, g["nomatch_qz"], Mq, N, nS)
        want = g["nomatch_s%d" % nS]
        assert np.array_equal(recs, want.view(recs.dtype).reshape(-1))
        assert nomatch > 0
    dense, nomatch_d, _ = orc.match_sweep(g["nomatch_pz"], Mp, g["nomatch_qz"], Mq, N)
    s1 = orc.match_sweep_sparse(g["nomatch_pz"], Mp, g["nomatch_qz"], Mq, N, 1)[0]
    assert np.array_equal(dense, g["nomatch_dense"].view(dense.dtype).reshape(-1))
    assert len(s1) == len(dense) and all(np.array_equal(s1[f], dense[f]) for f in ("ai", "bi", "start", "end")) and not s1["sparse"].any()


def test_segment_blocks_compose_to_the_whole_panel(orc):
    """orc_segment (build + -stats maxWithin over a block of sites from a checkpointed cursor — how the full-length configs[2]
    run is checked block by block, tests/test_gpu_z_c3_full.py): blocks chained through their (a, d) give the whole panel's
    bytes, final state and histogram (= the golden-pinned build_bitcols / max_within paths, and the reference where built),
    also from several threads at once"""
    from concurrent.futures import ThreadPoolExecutor
    for (M, N, kind, seed, cuts) in [(300, 400, 0, 9, (0, 1, 130, 399, 400)), (70, 150, 1, 4, (0, 64, 150)), (2, 9, 1, 3, (0, 4, 9)), (1500, 96, 0, 5, (0, 40, 96))]:
        bits = orc.synth_bitcols(M, N, seed=seed, kind=kind)
        o = orc.build_bitcols(bits, M, with_d=True, dump_sites=cuts)
        want_hist = orc.max_within_hist(o["yz"], M, N)
        if orc.ref() is not None:
            r = orc.ref_build_bitcols(bits, M, with_d=True)
            assert np.array_equal(o["yz"], r["yz"])
        jobs = [(cuts[i], cuts[i + 1], o["a_dump"][i], o["d_dump"][i]) for i in range(len(cuts) - 1)]       # every block from the whole run's own checkpoint
        with ThreadPoolExecutor(4) as ex:
            segs = list(ex.map(lambda j: orc.segment(bits[j[0]:j[1]], M, j[0], N, j[2], j[3]), jobs))
        for i, s in enumerate(segs):
            assert np.array_equal(s["a"], o["a_dump"][i + 1]) and np.array_equal(s["d"], o["d_dump"][i + 1])
        assert np.array_equal(np.concatenate([s["yz"] for s in segs]), o["yz"])
        assert np.array_equal(sum(s["hist"] for s in segs), want_hist)
        # a wrong start state is refused (sentinels), and a chained run (each block from the previous block's OUTPUT) agrees too
        a, d = np.arange(M, dtype=np.int32), np.zeros(M + 1, np.int32); d[0] = d[M] = 1
        tot = np.zeros(N + 2, np.int64)
        for (k0, k1, _, _) in jobs:
            s = orc.segment(bits[k0:k1], M, k0, N, a, d, yz_cap=len(o["yz"]) + M + 16)
            a, d = s["a"], s["d"]; tot += s["hist"]
        assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"]) and np.array_equal(tot, want_hist)
        with pytest.raises(AssertionError):
            orc.segment(bits[:1], M, 5, N, a, np.zeros(M + 1, np.int32))
