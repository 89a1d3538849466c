"""GPU (-m gpu): the BASELINE.json configurations pinned to the oracle at their own width.

configs[2] (100 000 haplotypes, build + -maxWithin): the exact option set bench.py times
(OPT_WITH_D | OPT_WITHIN_HIST | OPT_PACK3: skeleton chain + packed fill + histogram sweep + pack3)
over 32 768 sites: histogram, .pbwt bytes and final a/d against the oracle, then every site's a/d
checksum on a second pass.
configs[4] shape (1 000 000 haplotypes, build + -maxWithin + -matchDynamic with a 10 000-haplotype
query panel): 512 sites, histogram + packed bytes + final state, the records of a window of sites,
and the query sweep's records / no-match count / totals against the oracle.
One M > 2^20 case (the wide fallback chain: step2_kernel / step_kernel with 2048-position tiles).

(File name: these tests allocate and free tens of GB; they sort after the rest of the -m gpu suite.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def device_panel(eng, N, seed, kind=0):
    import torch
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=seed, kind=kind)
    eng.sync()
    return buf


def bench_pass(amd, eng, buf, N, step=8192):
    """the way bench.py drives the engine: consecutive pass_advance calls of `step` sites with 8 look-ahead columns"""
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    eng.pass_begin(N)
    for k in range(0, N, step):
        n = min(step, N - k)
        eng.pass_advance(buf.data_ptr() + k * eng.wpc * 4, n, min(n + 8, N - k), opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    return eng.get_hist(N + 1), eng.get_packed(), a, d


@pytest.mark.parametrize("team", ["0", "1", "onepass"])
def test_config2_bench_path_32768_sites(gpu_lib, orc, team, monkeypatch):
    amd = gpu_lib
    monkeypatch.setenv("PBWTAMD_TEAM", "1" if team == "1" else "0")           # "1": the team-persistent chain (one launch per batch, skel_team_kernel)
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if team == "onepass" else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")  # the one-launch round (skel_onepass_kernel)
    M, N = 100000, 32768
    eng = amd.Engine(M, batch_sites=512)
    buf = device_panel(eng, N, seed=0x5EED0001)
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    hist, yz, a, d = bench_pass(amd, eng, buf, N)
    assert np.array_equal(yz, o["yz"]), "packed PBWT (.pbwt payload) differs from the oracle"
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    want = orc.max_within_hist(o["yz"], M, N)[: N + 1]
    assert np.array_equal(hist, want), "maxWithin histogram differs at length %d" % int(np.argmax(hist != want))
    # every site's a[] and d[] (the checksum consumer makes the fill write the haplotype ids too)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]), "a[] differs at site %d" % int(np.argmax(ca != o["csum_a"]))
    assert np.array_equal(cd, o["csum_d"]), "d[] differs at site %d" % int(np.argmax(cd != o["csum_d"]))
    # the read side (-read x.pbwt -maxWithin -stats) at this width: same histogram from the packed panel
    assert np.array_equal(eng.max_within(o["yz"], N, mode="hist"), want)


def split_panel_and_queries(bits, M, Q, orc):
    """columns of an (M + Q)-haplotype panel -> (panel columns, query columns); M must be a multiple of 32"""
    N = bits.shape[0]
    pb = np.zeros((N, orc.wpc_for(M)), np.uint32)
    pb[:, : M // 32] = bits[:, : M // 32]
    nqw = (Q + 31) // 32
    qb = np.zeros((N, orc.wpc_for(Q)), np.uint32)
    qb[:, :nqw] = bits[:, M // 32: M // 32 + nqw]
    if Q % 32:
        qb[:, nqw - 1] &= np.uint32((1 << (Q % 32)) - 1)
    return pb, qb


@pytest.mark.parametrize("kind,Q", [(0, 1000), (1, 10000)])
def test_config4_shape_million_haplotypes(gpu_lib, orc, kind, Q):
    import torch
    amd = gpu_lib
    M, N = 1000000, 512
    bits = orc.synth_bitcols(M + Q, N, seed=77, kind=kind)
    pb, qb = split_panel_and_queries(bits, M, Q, orc)
    del bits
    o = orc.build_bitcols(pb, M, with_d=True)
    eng = amd.Engine(M, batch_sites=512)
    buf = torch.from_numpy(pb.view(np.int32)).cuda()
    hist, yz, a, d = bench_pass(amd, eng, buf, N, step=256)
    del buf
    assert np.array_equal(yz, o["yz"])
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    want = orc.max_within_hist(o["yz"], M, N)[: N + 1]
    assert np.array_equal(hist, want), "maxWithin histogram differs at length %d" % int(np.argmax(hist != want))
    # records of a window of sites in callback order (sweep_within_kernel<0/1> at this width)
    lo, hi = (300, 308) if kind == 0 else (300, 302)
    got = eng.max_within_range(o["yz"], N, lo, hi)
    exp = orc.max_within_range(o["yz"], M, N, lo, hi)
    assert len(got) == len(exp) and np.array_equal(got, exp)
    del got, exp
    # -matchDynamic: the query panel against the 1 M-wide panel
    q = orc.build_bitcols(qb, Q, with_d=False, want_csum=False)
    recs, nom, tot = eng.match_sweep(o["yz"], N, q["yz"], Q)
    wrecs, wnom, wtot = orc.match_sweep(o["yz"], M, q["yz"], Q, N)
    assert nom == wnom and tuple(tot) == tuple(wtot)
    assert len(recs) == len(wrecs) and np.array_equal(recs, wrecs)


@pytest.mark.parametrize("M,N,skel", [(1100000, 40, "1"), (2097152, 16, "1"), (2200000, 24, "1"), (4194000, 8, "1"), (2200000, 16, "0")])
def test_wider_than_2_20_haplotypes(gpu_lib, orc, M, N, skel, monkeypatch):
    """M > 2^20: the skeleton keeps its 512-position tiles with pair rows — the two-level scan on up to 2048 rows of pairs with 32 rows per
    scan workgroup (1.1 M; 2 097 152 = the last width of that form), 64 rows per workgroup up to the 4096 rows the engine takes (2.2 M,
    4 194 000 ~ 2^22) — and, with PBWTAMD_SKEL=0, the wide fallback chain (2048-position tiles, two sites per launch), which was all there
    was above 2^21 until round 3"""
    monkeypatch.setenv("PBWTAMD_SKEL", skel)
    amd = gpu_lib
    eng = amd.Engine(M, batch_sites=16)
    buf = device_panel(eng, N, seed=21)
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    b = eng.build(bits, with_d=True)
    assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"])
    hist, yz, a, d = bench_pass(amd, eng, buf, N, step=16)
    assert np.array_equal(yz, o["yz"]) and np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    if M <= 1200000:
        assert np.array_equal(hist, orc.max_within_hist(o["yz"], M, N)[: N + 1])
    else:
        # the oracle's matchMaximalWithin walks are quadratic in M on a panel's first sites (every haplotype still in one block): 3 minutes
        # at 2.2 M, hours at 4 M.  The histogram is compared with the OTHER chain's instead — skeleton against the two-site fallback:
        # independent (a, d) for every site through the same sweep, both pinned to the oracle's final state and bytes above
        monkeypatch.setenv("PBWTAMD_SKEL", "0" if skel == "1" else "1")
        other = amd.Engine(M, batch_sites=16)
        hist2, yz2, a2, d2 = bench_pass(amd, other, buf, N, step=16)
        other.close()
        monkeypatch.setenv("PBWTAMD_SKEL", skel)
        assert np.array_equal(yz2, o["yz"]) and np.array_equal(a2, o["aFend"]) and np.array_equal(d2, o["d_final"])
        assert np.array_equal(hist, hist2) and int(hist.sum()) > 0
    if M > 3000000:
        return                                     # (the read side is covered at 2.2 M)
    sw = eng.sweep_AD(o["yz"], N)
    s = orc.sweep_AD(o["yz"], M, N)
    assert np.array_equal(sw["csum_a"], s["csum_a"]) and np.array_equal(sw["csum_d"], s["csum_d"])


def test_bench_north_star_width_path(gpu_lib, orc):
    """bench.py's `north_star_width` measurement (1 M haplotypes, the bench option set, 8192-site advances) run on a
    short panel: the histogram total it reports and the bytes it packs are the oracle's — the published 1 M-wide number
    rests on verified output"""
    import sys
    import torch
    sys.path.insert(0, __import__("conftest").ROOT)
    import bench
    amd = gpu_lib
    sites, batch = 1024, 512
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    r = bench.north_star_width(torch, amd, torch.device("cuda", 0), opts, kind=0, sites=sites, batch=batch, step=768, want_hist=True)
    N = sites + batch
    bits = r["panel"].cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, 1000000, with_d=True, want_csum=False)
    want = orc.max_within_hist(o["yz"], 1000000, N)[: N + 1]
    assert np.array_equal(r["hist"], want) and r["within_reports_hist_total"] == int(want.sum())
    assert np.array_equal(r["packed"], o["yz"])
    assert r["roofline"]["sites_per_launch"] > 2.5 and 0 < r["roofline"]["frac"] < 1
