"""GPU (-m gpu): the HIP path through the C ABI against the oracle and the committed goldens.
Bit-exact everywhere: this is integer/index work."""
import os

import numpy as np
import pytest

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from conftest import GOLDEN, golden_panels, parse_pbwt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd(gpu_lib):
    assert gpu_lib.load_library().pbwtamd_device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return gpu_lib


def test_merge1_known_answer(amd, orc):
    M, N, aFstart, aFend, yz = parse_pbwt(os.path.join(GOLDEN, "merge1.pbwt"))
    g = np.load(os.path.join(GOLDEN, "merge1.npz"))
    eng = amd.Engine(M, batch_sites=4)
    sw = eng.sweep_AD(yz, N, aFstart, dump_sites=range(N + 1))
    assert np.array_equal(sw["a_dump"], g["sweep_a"])
    assert np.array_equal(sw["d_dump"], g["sweep_d"])
    recs = eng.max_within(yz, N, aFstart)
    assert np.array_equal(recs, g["within"])
    got = []
    eng.max_within(yz, N, aFstart, mode="callback", callback=lambda a, b, s, e: got.append((a, b, s, e)))
    assert got == [tuple(r) for r in g["within"].tolist()]
    lines = ["MATCH\t%d\t%d\t%d\t%d\t%d\n" % (a, b, s, e, e - s) for (a, b, s, e) in got if s != e]
    assert "".join(lines) == open(os.path.join(GOLDEN, "merge1.maxwithin.txt")).read()


@pytest.mark.parametrize("path", golden_panels(), ids=os.path.basename)
@pytest.mark.parametrize("batch", [7, 64])
def test_golden_panel(amd, orc, path, batch):
    g = np.load(path)
    M, N = int(g["M"]), int(g["N"])
    eng = amd.Engine(M, batch_sites=batch)
    # build with and without d: packed PBWT bytes, final a, final d
    b = eng.build(g["bits"], with_d=True)
    assert np.array_equal(b["yz"], g["yz"])
    assert np.array_equal(b["aFend"], g["aFend"])
    assert np.array_equal(b["dFend"], g["build_d"][N])
    bA = eng.build(g["bits"], with_d=False)
    assert np.array_equal(bA["yz"], g["yz"]) and np.array_equal(bA["aFend"], g["aFend"])
    # read-side sweep: every site's a and d
    sites = list(range(0, N + 1, max(1, N // 25))) + [N]
    sw = eng.sweep_AD(g["yz"], N, dump_sites=sites)
    for q, k in enumerate(sites):
        assert np.array_equal(sw["a_dump"][q], g["sweep_a"][k]), k
        assert np.array_equal(sw["d_dump"][q], g["sweep_d"][k]), k
    for k in range(N + 1):
        assert sw["csum_a"][k] == orc.checksum_i32(g["sweep_a"][k]), k
        assert sw["csum_d"][k] == orc.checksum_i32(g["sweep_d"][k]), k
        if k < N:
            assert sw["csum_y"][k] == orc.checksum_u8(g["sweep_y"][k]), k
    # maxWithin: records in callback order, and the -stats histogram
    assert np.array_equal(eng.max_within(g["yz"], N), g["within"])
    hist = eng.max_within(g["yz"], N, mode="hist")
    txt = "".join("%d\t%d\n" % (i, c) for i, c in enumerate(hist) if c)
    assert txt == g["hist_txt"].tobytes().decode()


@pytest.mark.parametrize("M,N,kind,batch", [(2, 9, 1, 4), (3, 50, 1, 16), (64, 64, 1, 64), (65, 33, 0, 8), (1025, 70, 0, 32),
                                          (2500, 300, 0, 128), (5000, 200, 1, 64), (20000, 150, 0, 50)])
def test_build_and_within_vs_oracle(amd, orc, M, N, kind, batch):
    bits = orc.synth_bitcols(M, N, seed=1000 + M, kind=kind)
    o = orc.build_bitcols(bits, M, with_d=True)
    eng = amd.Engine(M, batch_sites=batch)
    b = eng.build(bits, with_d=True)
    assert np.array_equal(b["yz"], o["yz"])
    assert np.array_equal(b["aFend"], o["aFend"])
    assert np.array_equal(b["dFend"], o["d_final"])
    s = orc.sweep_AD(o["yz"], M, N)
    sw = eng.sweep_AD(o["yz"], N)
    assert np.array_equal(sw["csum_a"], s["csum_a"])
    assert np.array_equal(sw["csum_d"], s["csum_d"])
    assert np.array_equal(sw["csum_y"][:N], s["csum_y"][:N])
    assert np.array_equal(eng.max_within(o["yz"], N, mode="hist"), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    if M * N <= 400000:
        assert np.array_equal(eng.max_within(o["yz"], N), orc.max_within(o["yz"], M, N))


@pytest.mark.parametrize("M,N,batch,avail_extra", [(3000, 301, 64, 2), (3000, 300, 64, 1), (700, 97, 10, 2), (300, 5, 4, 2), (1030, 64, 64, 2)])
def test_two_site_launch_paths(amd, orc, M, N, batch, avail_extra):
    """device pass over original-order columns: two-site launches (look-ahead of two columns), the
    single-site fallback (look-ahead of one), odd panel lengths and ragged batches; every site's a/d"""
    import torch
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=5, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM
    eng.pass_begin(N)
    k, step = 0, 2 * batch + 1                    # odd advance sizes exercise the re-prepare path
    while k < N:
        n = min(step, N - k)
        eng.pass_advance(buf.data_ptr() + k * eng.wpc * 4, n, min(n + avail_extra, N - k), opts)
        k += n
    eng.pass_end(opts)
    a, d = eng.get_state()
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    if avail_extra == 2:
        assert eng.chain_timing()[1] < N          # fewer launches than sites


def test_nonidentity_start_order(amd, orc):
    """aFstart other than the identity (a .pbwt written after a panel transform)"""
    M, N = 777, 90
    rng = np.random.default_rng(5)
    a0 = rng.permutation(M).astype(np.int32)
    bits = orc.synth_bitcols(M, N, seed=42, kind=0)
    o = orc.build_bitcols(bits, M, with_d=True, a0=a0)
    eng = amd.Engine(M, batch_sites=32)
    b = eng.build(bits, with_d=True, aFstart=a0)
    assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"])
    assert np.array_equal(eng.max_within(o["yz"], N, a0), orc.max_within(o["yz"], M, N, a0))


def test_degenerate_columns(amd, orc):
    """all-zero / all-one / single-carrier columns and alternating alleles (worst case for runs)"""
    M, N = 1500, 40
    hap = np.zeros((N, M), np.uint8)
    hap[1] = 1
    hap[2, 0] = 1
    hap[3, M - 1] = 1
    hap[4, ::2] = 1
    hap[5, 1::2] = 1
    hap[6:, :] = (np.random.default_rng(3).random((N - 6, M)) < 0.002)
    bits = orc.pack_bitcols(hap)
    o = orc.build_bitcols(bits, M, with_d=True)
    eng = amd.Engine(M, batch_sites=16)
    b = eng.build(bits, with_d=True)
    assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"])
    assert np.array_equal(eng.max_within(o["yz"], N), orc.max_within(o["yz"], M, N))


def test_pack3_codec_device(amd, orc):
    rng = np.random.default_rng(9)
    M, N = 70000, 12
    hap = np.zeros((N, M), np.uint8)
    hap[1] = 1
    hap[2] = rng.random(M) < 0.5
    hap[3] = rng.random(M) < 0.001
    hap[4, 100:64000] = 1                      # a run longer than 63488 is split (pbwtCore.c:247)
    hap[5, :2048] = 1
    hap[6:] = rng.random((N - 6, M)) < 0.05
    sb = orc.pack_bitcols(hap)
    want = np.concatenate([orc.pack3(hap[k]) for k in range(N)])
    eng = amd.Engine(M, batch_sites=5)
    assert np.array_equal(eng.pack3(sb), want)
    assert np.array_equal(eng.unpack3(want, N), sb)


def test_synth_generator_matches_oracle(amd, orc):
    import torch
    M, N = 3333, 70
    eng = amd.Engine(M, batch_sites=16)
    for kind in (0, 1):
        buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
        eng.synth_device(buf.data_ptr(), 5, N, seed=77, kind=kind)
        eng.sync()
        got = buf.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, orc.synth_bitcols(M, N, seed=77, kind=kind, k0=5))


def test_device_pass_api_with_graph(amd, orc):
    """device-resident columns, full-size batches (hipGraph path) + a ragged tail"""
    import torch
    M, N, B = 9000, 700, 256
    eng = amd.Engine(M, batch_sites=B)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=3, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    o = orc.build_bitcols(bits, M, with_d=True)
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    ms, n = eng.chain_timing()
    assert eng.chain_sites() == N and ms > 0
    if os.environ.get("PBWTAMD_SKEL", "1") == "0":
        assert n == N // 2                                        # the two-site chain
    else:
        assert n < N // 2                                         # 3 launches per 8 sites where the skeleton applies


@pytest.mark.parametrize("path", golden_panels(), ids=os.path.basename)
def test_match_sweep_golden(amd, orc, path):
    """matchSequencesSweep records, in the reference's callback order, for the held-out query split"""
    g = np.load(path)
    M, N, Mq = int(g["M"]), int(g["N"]), int(g["Mq"])
    eng = amd.Engine(M - Mq, batch_sites=37)
    recs, nomatch, tot = eng.match_sweep(g["pz"], N, g["qz"], Mq)
    assert np.array_equal(recs, g["qrecs"])
    _, o_nomatch, o_tot = orc.match_sweep(g["pz"], M - Mq, g["qz"], Mq, N)
    assert nomatch == o_nomatch and tuple(tot) == tuple(o_tot)


@pytest.mark.parametrize("Mp,Mq,N,kind,batch", [(1, 3, 20, 1, 8), (5, 1, 40, 1, 16), (700, 90, 300, 0, 64), (3000, 400, 500, 0, 128),
                                               (2100, 300, 260, 1, 100)])
def test_match_sweep_vs_oracle(amd, orc, Mp, Mq, N, kind, batch):
    bits = orc.synth_bitcols(Mp + Mq, N, seed=31 + Mp, kind=kind)
    hap = orc.unpack_bitcols(bits, Mp + Mq)
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qz = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    want, w_nomatch, w_tot = orc.match_sweep(pz, Mp, qz, Mq, N)
    eng = amd.Engine(Mp, batch_sites=batch)
    recs, nomatch, tot = eng.match_sweep(pz, N, qz, Mq)
    assert np.array_equal(recs, want)
    assert nomatch == w_nomatch and tuple(tot) == tuple(w_tot)
    got = []
    eng.match_sweep(pz, N, qz, Mq, callback=lambda a, b, s, e: got.append((a, b, s, e)))
    assert got == [tuple(r) for r in want.tolist()]


@pytest.mark.parametrize("Mp,Mq,N,kind,batch,popts", [(5, 3, 40, 1, 16, 0), (700, 90, 300, 0, 64, 0), (3000, 400, 501, 0, 128, 1), (20000, 64, 136, 1, 64, 1), (70001, 33, 96, 0, 32, 1),
                                                      (150600, 20, 48, 0, 16, 0)])
def test_match_sweep_stream_vs_oracle(amd, orc, Mp, Mq, N, kind, batch, popts):
    """matchSequencesSweep STREAMED (pbwtamd_match_sweep_stream): neither panel is packed or uploaded — the library asks a callback for the original-order
    bit columns of both panels a batch at a time (device pointers) and hands every batch's records to another.  Records, no-match count and totals equal
    the oracle's; with panel_opts the SAME pass also feeds the -stats histogram and pack3 (drained piece by piece: the pieces concatenate to the .pbwt
    payload), and a ragged last batch (N = 501) takes the fallback chains."""
    import torch
    bits = orc.synth_bitcols(Mp + Mq, N, seed=77 + Mp, kind=kind)
    hap = orc.unpack_bitcols(bits, Mp + Mq)
    pb = np.ascontiguousarray(orc.pack_bitcols(hap[:, :Mp])); qb = np.ascontiguousarray(orc.pack_bitcols(hap[:, Mp:]))
    op = orc.build_bitcols(pb, Mp, with_d=True)
    qz = orc.build_bitcols(qb, Mq, with_d=False)["yz"]
    want, w_nomatch, w_tot = orc.match_sweep(op["yz"], Mp, qz, Mq, N)
    eng = amd.Engine(Mp, batch_sites=batch)
    dp = torch.from_numpy(pb.view(np.int32)).cuda(); dq = torch.from_numpy(qb.view(np.int32)).cuda()
    torch.cuda.synchronize()
    assert pb.shape[1] == eng.wpc and qb.shape[1] == amd.wpc_for(Mq)
    asked, pieces = [], []
    opts = (amd.OPT_WITHIN_HIST | amd.OPT_PACK3) if popts else 0

    def cols(site0, ncols):
        asked.append((site0, ncols))
        assert site0 + ncols <= N
        if popts and site0:
            pieces.append(eng.drain_packed().copy())        # the bytes of the batches before this one leave the engine here
        return dp.data_ptr() + site0 * eng.wpc * 4, dq.data_ptr() + site0 * qb.shape[1] * 4

    recs, nomatch, tot = eng.match_sweep_stream(N, Mq, cols, panel_opts=opts)
    assert [a for a, _ in asked] == list(range(0, N, batch))
    got = np.zeros(len(recs), amd.MATCH_DTYPE)
    for f in ("ai", "bi", "start", "end"):
        got[f] = recs[f]
    assert np.all(recs["sparse"] == 0)
    assert np.array_equal(got, want)
    assert nomatch == w_nomatch and tuple(tot) == tuple(w_tot)
    if popts:
        pieces.append(eng.drain_packed().copy())
        assert np.array_equal(np.concatenate(pieces), op["yz"]), "the drained pieces are not the .pbwt payload"
        assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(op["yz"], Mp, N)[: N + 1])
    chunks = []
    eng.match_sweep_stream(N, Mq, cols, on_records=lambda a: chunks.append(len(a)))     # records delivered batch by batch, nothing kept by the library
    assert sum(chunks) == len(want)


def test_haplotypes_and_y_dump(amd, orc):
    M, N = 1234, 77
    bits = orc.synth_bitcols(M, N, seed=8, kind=0)
    o = orc.build_bitcols(bits, M, with_d=False)
    eng = amd.Engine(M, batch_sites=20)
    assert np.array_equal(eng.haplotypes(o["yz"], N), orc.unpack_bitcols(bits, M))
    sites = [0, 1, 19, 20, 21, 76]
    sw = eng.sweep_AD(o["yz"], N, dump_sites=sites)
    s = orc.sweep_AD(o["yz"], M, N, dump_sites=sites)
    assert np.array_equal(sw["y_dump"], s["y_dump"]) and np.array_equal(sw["d_dump"], s["d_dump"]) and np.array_equal(sw["a_dump"], s["a_dump"])


@pytest.mark.parametrize("pair1024", ["0", "1"])
def test_large_panel_1024_position_tiles(amd, orc, pair1024, monkeypatch):
    """M > 262144 uses 1024-position tiles: single-site launches (4 positions per thread) by default,
    two-site launches with 1024-thread workgroups when PBWTAMD_PAIR1024=1"""
    import torch
    monkeypatch.setenv("PBWTAMD_PAIR1024", pair1024)
    M, N = 300000, 41
    eng = amd.Engine(M, batch_sites=16)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=2, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    b = eng.build(bits, with_d=True)
    assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"])
    assert np.array_equal(eng.max_within(o["yz"], N, mode="hist"), orc.max_within_hist(o["yz"], M, N)[: N + 1])


@pytest.mark.parametrize("M,N,kind,L,batch", [(8, 10, 1, 2, 4), (70, 150, 1, 5, 16), (300, 400, 0, 30, 64), (300, 400, 0, 100, 33), (2500, 300, 0, 80, 128)])
def test_long_within_vs_oracle(amd, orc, M, N, kind, L, batch):
    bits = orc.synth_bitcols(M, N, seed=77 + M, kind=kind)
    yz = orc.build_bitcols(bits, M, with_d=False)["yz"]
    eng = amd.Engine(M, batch_sites=batch)
    assert np.array_equal(eng.long_within(yz, N, L), orc.long_within(yz, M, N, L))


def test_mode_switches_inside_a_pass(amd, orc):
    """alternate look-ahead of one / two columns between advances: the pass switches between
    single-site and two-site launches (re-deriving tags and summaries at each switch)"""
    import torch
    M, N = 2000, 300
    eng = amd.Engine(M, batch_sites=32)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=21, kind=0)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST
    eng.pass_begin(N)
    k, i = 0, 0
    for n in [40, 7, 64, 33, 1, 2, 90, 63]:
        n = min(n, N - k)
        eng.pass_advance(buf.data_ptr() + k * eng.wpc * 4, n, min(n + 1 + (i & 1), N - k), opts)
        k += n; i += 1
    assert k == N
    eng.pass_end(opts)
    a, d = eng.get_state()
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])


def test_degenerate_shapes(amd, orc):
    """empty panel (N = 0), a single site, a single haplotype, width not a multiple of 32/64/256"""
    # N = 0: nothing to build; the cursor stays at its start order with the initial sentinels
    eng = amd.Engine(5, batch_sites=4)
    b = eng.build(np.zeros((0, eng.wpc), np.uint32), with_d=True)
    assert len(b["yz"]) == 0 and b["aFend"].tolist() == [0, 1, 2, 3, 4] and b["dFend"].tolist() == [1, 0, 0, 0, 0, 1]
    sw = eng.sweep_AD(np.zeros(0, np.uint8), 0)
    assert sw["csum_a"][0] == orc.checksum_i32(np.arange(5)) and sw["csum_d"][0] == orc.checksum_i32(np.array([1, 0, 0, 0, 0, 1]))
    # N = 1 and M = 1
    for M, N in [(5, 1), (1, 7), (1, 1), (33, 3), (257, 2)]:
        bits = orc.synth_bitcols(M, N, seed=M * 10 + N, kind=1)
        o = orc.build_bitcols(bits, M, with_d=True)
        eng = amd.Engine(M, batch_sites=4)
        b = eng.build(bits, with_d=True)
        assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"]), (M, N)
        assert np.array_equal(eng.haplotypes(o["yz"], N), orc.unpack_bitcols(bits, M))
        if M >= 2:
            assert np.array_equal(eng.max_within(o["yz"], N), orc.max_within(o["yz"], M, N))
    # maxWithin refuses a single haplotype (the reference would read y[-1])
    eng1 = amd.Engine(1, batch_sites=4)
    with pytest.raises(amd.PbwtAmdError, match="at least 2 haplotypes"):
        eng1.max_within(orc.pack3(np.zeros(1, np.uint8)), 1)


def test_malformed_packed_panel_is_rejected(amd, orc):
    M, N = 100, 10
    bits = orc.synth_bitcols(M, N, seed=1, kind=1)
    yz = orc.build_bitcols(bits, M, with_d=False)["yz"]
    eng = amd.Engine(M, batch_sites=4)
    with pytest.raises(amd.PbwtAmdError, match="decodes to"):
        eng.max_within(yz[:-3], N, mode="hist")          # truncated
    with pytest.raises(amd.PbwtAmdError, match="decodes to"):
        eng.max_within(yz, N + 1, mode="hist")           # wrong N
    # a crafted file whose run lengths still total M*N but with a run straddling a column boundary: the two runs
    # either side of the boundary between columns 3 and 4 are merged into one (same allele) or the boundary is moved
    # by shifting one position from the last run of column 3 to the first run of column 4
    import oracle
    starts = [0]
    for k in range(N):
        _, used, _ = oracle.unpack3(yz[starts[-1]:], M)
        starts.append(starts[-1] + used)
    bad = yz.copy()
    i, j = starts[4] - 1, starts[4]                      # last byte of column 3, first byte of column 4
    if (bad[i] & 0x7f) > 1 and (bad[j] & 0x7f) < 63:
        bad[i] -= 1; bad[j] += 1                         # column 3 is one position short, column 4 one long
        with pytest.raises(amd.PbwtAmdError, match="malformed packed panel"):
            eng.max_within(bad, N, mode="hist")
        with pytest.raises(amd.PbwtAmdError, match="malformed packed panel"):
            eng.haplotypes(bad, N)
    # the start order drives device gathers: anything but a permutation of [0, M) is refused
    with pytest.raises(amd.PbwtAmdError, match="not a permutation"):
        eng.max_within(yz, N, aFstart=np.zeros(M, np.int32), mode="hist")
    with pytest.raises(amd.PbwtAmdError, match="not a permutation"):
        eng.build(bits, aFstart=np.arange(1, M + 1, dtype=np.int32))
    assert np.array_equal(eng.max_within(yz, N, mode="hist"), orc.max_within_hist(yz, M, N)[: N + 1])   # the engine is still usable


@pytest.mark.parametrize("skel", ["1", "0", "onepass", "scan"])
@pytest.mark.parametrize("M,N,batch,kind", [(3000, 264, 64, 0), (1024, 130, 32, 1), (1025, 96, 24, 0), (70001, 80, 40, 0),
                                            (300000, 40, 16, 0), (600100, 24, 8, 0), (1500, 41, 8, 1), (5, 64, 16, 1), (1, 16, 8, 0),
                                            (150600, 40, 16, 0), (524288, 24, 8, 1),    # pair rows with the one-level scan: 148 rows (odd tile count), 512 rows
                                            (9000, 136, 64, 0), (12288, 96, 32, 1), (8193, 72, 24, 0),    # two launches per round on 512-position tiles (17-24 tiles)
                                            (525000, 16, 8, 0), (1048576, 16, 8, 1)])   # the local scan launch (skel_k2_local_kernel) at both ends of its range: 513 and 1 024 scan rows
def test_both_chains_every_site(amd, orc, skel, M, N, batch, kind, monkeypatch):
    """the two chain implementations — skeleton (8-bit radix step every 8 sites, K1/K2/K3, with the
    seven states between filled by batched single-site kernels) and the two-site chain — against the
    oracle at EVERY site (checksums of a and d), plus the consumers fed from those states (hist, pack3)"""
    import torch
    # "onepass": the skeleton with ONE launch per round of 8 sites (skel_onepass_kernel: totals precomputed, two-level look-back inside the launch) instead of
    # three (two below 12 289 haplotypes); it takes every width up to 1 024 tiles of 512 positions
    # (its look-back form, up to 1 024 tiles here: PBWTAMD_ONEPASS_MAXW); "scan": its SCANNER form (round 6: scanner and aggregator workgroups in front of the tiles do the
    # scan over the tiles inside the launch, XCD-local groups, tiles in dispatch order; measured slower than three launches, so opt-in) at every width
    if skel == "onepass" and M > 524288:
        pytest.skip("the one-launch round's look-back form takes up to 1 024 tiles")
    monkeypatch.setenv("PBWTAMD_SKEL", "0" if skel == "0" else "1")
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if skel in ("onepass", "scan") else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")
    if skel == "scan":
        monkeypatch.setenv("PBWTAMD_ONEPASS_SCAN", "1"); monkeypatch.setenv("PBWTAMD_ONEPASS_SCAN_MIN", "0")
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=1000 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | (amd.OPT_WITHIN_HIST if M > 1 else 0)
    if M == 1:                                 # the reference's maxWithin reads y[-1] for a single haplotype: refused
        eng.pass_begin(N)
        with pytest.raises(amd.PbwtAmdError):
            eng.pass_advance(buf.data_ptr(), N, N, opts | amd.OPT_WITHIN_HIST)
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]), "a[] differs first at site %d" % int(np.argmax(ca != o["csum_a"]))
    assert np.array_equal(cd, o["csum_d"]), "d[] differs first at site %d" % int(np.argmax(cd != o["csum_d"]))
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    if M > 1:
        assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    n = eng.chain_timing()[1]
    if skel == "1" and N >= batch and batch % 8 == 0:
        assert n < (N + 1) // 2                # at least one batch went through the skeleton (3 launches per 8 sites)
    if skel in ("onepass", "scan") and batch % 8 == 0:
        assert n <= N // 8 + 2 * (N % batch) + 2, "the one-launch round did not take the batches (%d chain launches for %d sites)" % (n, N)
    b = eng.build(bits, with_d=True)           # host entry point: pack3 stream + records sink
    assert np.array_equal(b["yz"], o["yz"]) and np.array_equal(b["dFend"], o["d_final"])
    if 1 < M <= 3000:
        assert np.array_equal(eng.max_within(o["yz"], N, mode="records"), orc.max_within(o["yz"], M, N))


@pytest.mark.parametrize("M,N,batch,kind,K", [(70001, 80, 40, 0, 0), (100000, 136, 64, 0, 0), (100000, 72, 24, 1, 7), (20000, 136, 64, 1, 79), (139000, 40, 16, 0, 128),
                                              (30000, 520, 256, 0, 33), (131072, 24, 8, 1, 64)])
def test_team_chain_every_site(amd, orc, M, N, batch, kind, K, monkeypatch):
    """the third chain form (PBWTAMD_TEAM=1, skel_team_kernel): all rounds of a batch in ONE launch, hist / scan / rank separated by flag-word barriers
    among the workgroups of one XCD instead of by kernel boundaries — against the oracle at EVERY site (checksums of a and d), plus the consumers
    fed from those states (histogram, .pbwt bytes), with one tile per member, several tiles per member (K = 7) and odd team sizes."""
    import torch
    monkeypatch.setenv("PBWTAMD_TEAM", "1")
    monkeypatch.setenv("PBWTAMD_ONEPASS", "0")             # (the team form replaces the three-launch round; the one-launch round is a form of its own)
    if K:
        monkeypatch.setenv("PBWTAMD_TEAM_K", str(K))
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=1700 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST
    eng.pass_begin(N)
    half = (N // 2) // 8 * 8                   # two advances: the second starts from a carried cursor
    eng.pass_advance(buf.data_ptr(), half, N, opts)
    eng.pass_advance(buf.data_ptr() + half * eng.wpc * 4, N - half, N - half, opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]), "a[] differs first at site %d" % int(np.argmax(ca != o["csum_a"]))
    assert np.array_equal(cd, o["csum_d"]), "d[] differs first at site %d" % int(np.argmax(cd != o["csum_d"]))
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    ms, n = eng.chain_timing()
    assert n <= (N + 7) // 8 + 2 * (N % batch) + 8, "the team form did not take the batches (%d chain launches for %d sites)" % (n, N)
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3          # the bench option set: packed fill, 16-bit hand-off, pack3 from the sweep's columns
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])


@pytest.mark.parametrize("M,N,batch,kind", [(70001, 80, 40, 0), (100000, 72, 24, 1)])
def test_team_that_does_not_fill_falls_back(amd, orc, M, N, batch, kind, monkeypatch, capfd):
    """a team of the team-persistent chain that does not fill (a device or partition with fewer XCDs than panels, no room beside other kernels) must not
    fail the pass or, worse, leave a panel unadvanced without a word: team_batch reads the tickets taken per XCD behind every launch; PBWTAMD_TEAM_FAIL_ONCE=1 makes
    that check fail once — the batch is run again with three launches per round, the rest of the pass stays there, the states equal the oracle's at every site"""
    import torch
    monkeypatch.setenv("PBWTAMD_TEAM", "1"); monkeypatch.setenv("PBWTAMD_ONEPASS", "0"); monkeypatch.setenv("PBWTAMD_TEAM_FAIL_ONCE", "1")
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=1800 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    ms, n = eng.chain_timing()
    assert n >= 3 * ((N // batch) * (batch // 8) - batch // 8), "the batches after the failed one did not take three launches per round (%d chain launches)" % n


@pytest.mark.parametrize("M,N,batch,kind", [(100000, 72, 24, 1), (30000, 136, 64, 0), (300000, 24, 8, 0)])
def test_onepass_round_in_dispatch_order(amd, orc, M, N, batch, kind, monkeypatch):
    """PBWTAMD_ONEPASS_ORDERED=1: the one-launch round with tile = workgroup index (what the library uses where several chains run at once: a tile then waits only for
    workgroups dispatched before it) — every site's a / d, histogram and .pbwt bytes against the oracle"""
    import torch
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024"); monkeypatch.setenv("PBWTAMD_ONEPASS_ORDERED", "1")
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=4100 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])


@pytest.mark.parametrize("folders,both", [("0", "0"), ("1", "0"), ("1", "1")])
@pytest.mark.parametrize("M,N,batch,kind", [(100000, 72, 24, 1), (40000, 136, 64, 0), (131072, 24, 8, 0), (3000, 40, 40, 1)])
def test_onepass_look_back_forms(amd, orc, M, N, batch, kind, folders, both, monkeypatch):
    """the one-launch round's look-back in its three forms — the group's last tile folding the group (PBWTAMD_ONEPASS_FOLDERS=0), a folder workgroup per group
    with the tiles polling level 1 then level 2 (FOLDERS=1, BOTH=0), and both levels in one round trip (BOTH=1) — on 256- and 512-position tiles, one tile
    and 256 tiles included: every site's a / d, histogram and .pbwt bytes against the oracle"""
    import torch
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1"); monkeypatch.setenv("PBWTAMD_ONEPASS_FOLDERS", folders); monkeypatch.setenv("PBWTAMD_ONEPASS_BOTH", both)
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=5200 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
    ca, cd, _ = eng.get_checksums(0, N + 1)
    assert np.array_equal(ca, o["csum_a"]) and np.array_equal(cd, o["csum_d"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])
    assert eng.chain_timing()[1] <= N // 8 + 2 * (N % batch) + 2, "the one-launch round did not take the batches"


@pytest.mark.parametrize("alone", ["0", "1"])
@pytest.mark.parametrize("M,N,batch,kind", [(100000, 72, 24, 1), (9000, 136, 64, 0), (600100, 24, 8, 0), (1, 16, 8, 0)])
def test_pack3_without_the_histogram(amd, orc, M, N, batch, kind, alone, monkeypatch):
    """OPT_PACK3 without OPT_WITHIN_HIST (a plain build + write of the .pbwt columns): by default such a pass takes the packed fill and the sweep's sorted
    columns (the histogram comes along unasked), PBWTAMD_PACK3_ALONE=1 the table fill — the .pbwt bytes and the final state against the oracle either way"""
    import torch
    monkeypatch.setenv("PBWTAMD_PACK3_ALONE", alone)
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=6300 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_PACK3
    eng.pass_begin(N); eng.pass_advance(buf.data_ptr(), N, N, opts); eng.pass_end(opts)
    assert np.array_equal(eng.get_packed(), o["yz"])
    a, d = eng.get_state()
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])


@pytest.mark.parametrize("packed", ["1", "0", "onepass"])
@pytest.mark.parametrize("M,N,batch,kind", [(3000, 264, 64, 0), (1025, 96, 24, 0), (70001, 80, 40, 1), (2, 40, 8, 1), (300000, 24, 8, 0),
                                            (600100, 24, 8, 1), (150600, 32, 16, 1), (139300, 24, 8, 0),      # 600 100: pair rows with an odd number of tiles (1173); 150 600 / 139 300: pair rows, one-level scan (the narrowest: 137 rows)
                                            (525000, 16, 8, 1), (1048576, 16, 8, 0)])   # 513 and 1 024 scan rows: the first and the last width of the local scan launch (aggx folded by rank and both fills)
def test_histogram_and_pack3_consumers_without_ids(amd, orc, packed, M, N, batch, kind, monkeypatch):
    """the bench configuration (divergence + maxWithin histogram + pack3, no per-site checksums): on the
    skeleton path the fill then writes d | y << 31 and no haplotype ids, the sweep reads that and emits
    the bit columns pack3 encodes (PBWTAMD_NO_PACKED_FILL=1: the unpacked form).  Histogram, .pbwt bytes
    and final state against the oracle."""
    import torch
    if packed == "0":
        monkeypatch.setenv("PBWTAMD_NO_PACKED_FILL", "1")
    if packed == "onepass" and M > 524288:
        pytest.skip("the one-launch round takes up to 1 024 tiles")
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if packed == "onepass" else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")     # the packed consumers behind the one-launch round (no pair rows at any width)
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=2000 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    eng.pass_begin(N)
    half = (N // 2) // 8 * 8                   # two advances: the second starts from a carried cursor
    eng.pass_advance(buf.data_ptr(), half, N, opts)
    eng.pass_advance(buf.data_ptr() + half * eng.wpc * 4, N - half, N - half, opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])


@pytest.mark.parametrize("form", ["seq", "seq_esc", "seq32", "table", "onepass", "onepass_table"] + (["fused", "fused_always", "yc"] if os.environ.get("PBWTAMD_MEASURE_BUILD") else []))
@pytest.mark.parametrize("M,N,batch,kind", [(3000, 264, 64, 0), (1025, 96, 24, 1), (70001, 80, 40, 1), (2, 40, 8, 1), (257, 64, 64, 1), (300000, 24, 8, 0),
                                            (600100, 24, 8, 1), (150600, 32, 16, 1), (9000, 72, 24, 0), (56001, 48, 16, 1), (511, 40, 8, 1), (512, 40, 8, 0),
                                            (100000, 136, 64, 0), (1000003, 16, 8, 0)])
def test_packed_fill_every_position(amd, orc, form, M, N, batch, kind, monkeypatch):
    """the packed fill (slots hold d | y << 31, the bench option set) checked at EVERY position of EVERY site: per-site checksums of d and y taken
    from the packed slots (PBWTAMD_PACKED_CHECKSUM=1) against the oracle's — for the sequential tile-local form (skel_fillseq_kernel,
    PBWTAMD_FILL_SEQ=1, the default; with and without the fused first step of matchMaximalWithin) and the table form (skel_fill_kernel).  iid panels put every 8-bit key into every tile; widths cover
    256- and 512-position tiles, ragged last tiles, pair rows (odd and even tile counts) and the two-launch round."""
    import torch
    # seq: sequential fill + streaming sweep (the shipped path); table: skel_fill_kernel.  Measurement builds (PBWTAMD_MEASURE_BUILD=1) also run the fused
    # forms — the sequential fill deciding the first step of the -stats sweep and emitting the bit columns, the rest through sweep_resid_kernel; a panel
    # that leaves more than 10 % undecided (iid) switches back after its first batches, fused_always never does — which passed here and measured slower
    # seq: the 16-bit hand-off (L | y << 15 slots, the shipped path); seq_esc: the same with lengths from 3 on escaping to the 32-bit slot (what a match of
    # 32 767 sites or more does in production: here most positions take that path); seq32: the d | y << 31 slots (PBWTAMD_P16=0)
    if form.startswith("onepass") and M > 524288:
        pytest.skip("the one-launch round takes up to 1 024 tiles")
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if form.startswith("onepass") else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")     # both fills behind the one-launch round's tables
    monkeypatch.setenv("PBWTAMD_P16", "0" if form == "seq32" else "1")
    monkeypatch.setenv("PBWTAMD_P16_CLIP", "3" if form == "seq_esc" else "32767")
    monkeypatch.setenv("PBWTAMD_FILL_SEQ", "0" if form in ("table", "onepass_table") else "1")
    monkeypatch.setenv("PBWTAMD_FILL_FUSE", "1" if form.startswith("fused") else "0")
    monkeypatch.setenv("PBWTAMD_FILL_YC", "1" if form == "yc" else "0")        # yc: the fill emits the sorted allele columns, the sweep reads them first
    if form == "fused_always":
        monkeypatch.setenv("PBWTAMD_FUSE_MAX_FLAGGED", "2")
    monkeypatch.setenv("PBWTAMD_PACKED_CHECKSUM", "1")
    eng = amd.Engine(M, batch_sites=batch)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=3000 + M, kind=kind)
    eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    sw = orc.sweep_AD(o["yz"], M, N)
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3 | amd.OPT_CHECKSUM
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    _, cd, cy = eng.get_checksums(0, N)
    full = N // batch * batch if batch % 8 == 0 else 0          # sites of skeleton batches (packed slots); the ragged tail runs the two-site chain
    assert np.array_equal(cd[:N], o["csum_d"][:N]), "d[] differs first at site %d" % int(np.argmax(cd[:N] != o["csum_d"][:N]))
    assert np.array_equal(cy[:N], sw["csum_y"][:N]), "y[] differs first at site %d" % int(np.argmax(cy[:N] != sw["csum_y"][:N]))
    assert full >= 0
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])


def test_p16_matches_longer_than_32766_sites(amd, orc, monkeypatch):
    """the 16-bit hand-off at its own escape threshold: a panel with duplicated haplotypes over 34 000 sites — their divergences stay at 0 for the
    whole panel, so from site 32 766 on those positions overflow the 15-bit length and take the escape (d from the 32-bit slot).  Every position
    of every site (checksums from the slots), the histogram and the .pbwt bytes against the oracle."""
    import torch
    M, N, batch = 700, 34000, 512
    hap = orc.unpack_bitcols(orc.synth_bitcols(M, N, seed=77, kind=1), M)
    hap[:, 1] = hap[:, 0]; hap[:, 301] = hap[:, 300]; hap[:, 699] = hap[:, 5]; hap[:, 6] = hap[:, 5]
    hap[20000:, 400] = hap[20000:, 401]                      # (a match that starts mid-panel: no overflow before the end)
    bits = np.ascontiguousarray(orc.pack_bitcols(hap))
    o = orc.build_bitcols(bits, M, with_d=True)
    sw = orc.sweep_AD(o["yz"], M, N)
    monkeypatch.setenv("PBWTAMD_PACKED_CHECKSUM", "1")
    monkeypatch.setenv("PBWTAMD_P16", "1")                   # (the default at every width; pinned here so that this test keeps testing the 16-bit hand-off if that changes)
    eng = amd.Engine(M, batch_sites=batch)
    assert bits.shape[1] == eng.wpc
    buf = torch.from_numpy(bits.view(np.int32)).cuda()
    torch.cuda.synchronize()
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3 | amd.OPT_CHECKSUM
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    _, cd, cy = eng.get_checksums(0, N)
    assert np.array_equal(cd[:N], o["csum_d"][:N]), "d[] differs first at site %d" % int(np.argmax(cd[:N] != o["csum_d"][:N]))
    assert np.array_equal(cy[:N], sw["csum_y"][:N]), "y[] differs first at site %d" % int(np.argmax(cy[:N] != sw["csum_y"][:N]))
    assert np.array_equal(eng.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    assert np.array_equal(eng.get_packed(), o["yz"])


@pytest.mark.parametrize("Mp,Mq,N,kind,nS,batch", [(8, 3, 10, 1, 2, 4), (50, 7, 61, 0, 3, 16), (200, 20, 150, 0, 4, 32), (64, 10, 33, 1, 5, 10),
                                                  (300, 25, 100, 0, 2, 512), (40, 6, 50, 1, 1, 16), (3000, 50, 200, 0, 4, 64), (2500, 30, 130, 1, 8, 128)])
def test_match_sweep_sparse_vs_oracle(amd, orc, Mp, Mq, N, kind, nS, batch):
    """matchSequencesSweepSparse (pbwtMatch.c:501-602): records incl. the isSparse flag, in callback order, totals"""
    bits = orc.synth_bitcols(Mp + Mq, N, seed=Mp * 7 + N, kind=kind)
    hap = orc.unpack_bitcols(bits, Mp + Mq)
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qz = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    want, nomatch, tot = orc.match_sweep_sparse(pz, Mp, qz, Mq, N, nS)
    eng = amd.Engine(Mp, batch_sites=batch)
    got, gn, gt = eng.match_sweep_sparse(pz, N, qz, Mq, nS)
    assert np.array_equal(got, want) and gn == nomatch and tuple(gt) == tuple(tot)
    if Mp <= 64:                               # callback form delivers the same stream
        seen = []
        eng.match_sweep_sparse(pz, N, qz, Mq, nS, callback=lambda a, b, s, e, sp: seen.append((a, b, s, e, sp)))
        assert seen == [tuple(r) for r in want.tolist()]


@pytest.mark.parametrize("Mp,Mq,N,rare,nS,batch", [(6000, 40, 160, 0.004, 0, 64), (20000, 64, 96, 0.0007, 0, 32), (9000, 30, 120, 0.002, 3, 64),
                                                    (70000, 48, 64, 0.0002, 0, 64), (5000, 32, 100, 0.5, 2, 32)])
def test_match_sweep_long_walks_block_skipping(amd, orc, Mp, Mq, N, rare, nS, batch, monkeypatch):
    """reportAndUpdate's walks (pbwtMatch.c:452-499) over thousands of positions: a panel in which one allele is rare (some sites
    carry none of it: the no-match branch), queries that carry it often.  The walks then skip whole blocks of 256 positions
    through the {max d, alleles present} summaries (qs_blocksum_kernel) — forwards for the scan for an equally long match,
    backwards for the widening, forwards again for the lowest candidate of a skipped stretch; records, no-match events and
    totals are the oracle's, and identical with the summaries switched off."""
    rng = np.random.default_rng(Mp + N)
    hap = np.zeros((N, Mp + Mq), np.uint8)
    p_site = np.where(rng.random(N) < 0.15, 0.0, rare)                                  # 15 % of the sites: nobody in the panel carries a 1
    hap[:, :Mp] = rng.random((N, Mp)) < p_site[:, None]
    flip = rng.random(N) < 0.3
    hap[flip, :Mp] ^= 1                                                                   # ... or the rare allele is the 0
    hap[:, Mp:] = rng.random((N, Mq)) < 0.5
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qz = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    want, nomatch, tot = orc.match_sweep_sparse(pz, Mp, qz, Mq, N, nS)
    eng = amd.Engine(Mp, batch_sites=batch)
    got, gn, gt = eng.match_sweep_sparse(pz, N, qz, Mq, nS)
    assert gn == nomatch and tuple(gt) == tuple(tot)
    assert len(got) == len(want)
    for f in ("ai", "bi", "start", "end", "sparse"):
        assert np.array_equal(got[f], want[f]), f
    if rare < 0.5:
        assert nomatch > 0
    ev = eng.nomatch_events()
    import subprocess, sys, tempfile                                                      # the switch is read once per process: a fresh interpreter for the A/B run
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), pz=pz, qz=qz)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import pbwt_amd as amd; z = np.load(%r); e = amd.Engine(%d, batch_sites=%d); "
                "r, n, t = e.match_sweep_sparse(z['pz'], %d, z['qz'], %d, %d); np.savez(%r, r=r, n=n, t=np.array(t), ev=e.nomatch_events())"
                % (ROOT_DIR, os.path.join(td, "in.npz"), Mp, batch, N, Mq, nS, os.path.join(td, "out.npz")))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PBWTAMD_QS_BLOCKS="0"), timeout=600)
        o = np.load(os.path.join(td, "out.npz"))
        assert np.array_equal(o["r"], got) and int(o["n"]) == gn and np.array_equal(o["ev"], ev)


def test_no_match_events_beyond_the_cap_keep_the_log_order(amd, orc):
    """matchSequencesSweep logs "no match to query jj value x at site k" (pbwtMatch.c:405-410) for every event, in site order; the engine keeps
    the first 65 536 for the caller's log and counts the rest.  They have to be the FIRST 65 536 in log order — site, then query rank — not
    whichever arrived first: a panel without a single 1 against queries of all 1s gives an event per (site, query), 90 000 here over five batches."""
    Mp, Mq, N = 16, 300, 300
    pz = orc.build_bitcols(orc.pack_bitcols(np.zeros((N, Mp), np.uint8)), Mp, with_d=False)["yz"]
    qz = orc.build_bitcols(orc.pack_bitcols(np.ones((N, Mq), np.uint8)), Mq, with_d=False)["yz"]
    want, w_nomatch, w_tot = orc.match_sweep(pz, Mp, qz, Mq, N)
    assert w_nomatch == N * Mq
    eng = amd.Engine(Mp, batch_sites=64)
    recs, nomatch, tot = eng.match_sweep(pz, N, qz, Mq)
    assert nomatch == w_nomatch and tuple(tot) == tuple(w_tot) and len(recs) == len(want)
    ev = eng.nomatch_events()
    assert ev.shape == (65536, 4)
    idx = np.arange(65536)
    assert np.array_equal(ev[:, 2], idx // Mq), "sites of the kept events are not the first ones in log order"
    assert np.array_equal(ev[:, 0], idx % Mq) and np.all(ev[:, 1] == 1) and np.all(ev[:, 3] == 0)   # identical queries keep their original order in the query PBWT


@pytest.mark.parametrize("team", ["0", "1", "onepass", "auto"])
@pytest.mark.parametrize("P,M,N,B", [(3, 5000, 602, 256), (4, 30000, 520, 256), (2, 100000, 264, 128), (2, 600100, 24, 8), (8, 60000, 264, 128), (11, 20000, 136, 64)])
def test_many_panels_per_launch(amd, orc, P, M, N, B, team, monkeypatch):
    """pbwtamd_pass_advance_many: P independent panels (chromosomes) of one width advance through fused chain launches (grid.y = panel;
    two launches per round at 5 000 haplotypes, three at 30 000 / 100 000; the 600 100-wide case and the ragged tails take the per-engine
    fallback).  Every panel's .pbwt bytes, final a / d and -stats histogram equal the oracle's for that panel alone."""
    import torch
    # team = "1": panel p on XCD p, all rounds of a batch in one launch (skel_team_kernel), eight panels at a time (the 11-panel case: 8 + 3)
    if team == "1" and (M <= 12288 or M > 139000):
        pytest.skip("the team form takes the widths of the three-launch round without pair rows")
    if team == "auto":                                      # the shipped choice: the one-launch round with grid.y = panel, the team form for whole sets of eight panels
        if P % 8:
            pytest.skip("as the onepass case")
        monkeypatch.delenv("PBWTAMD_TEAM", raising=False); monkeypatch.delenv("PBWTAMD_ONEPASS", raising=False); monkeypatch.delenv("PBWTAMD_ONEPASS_MAXW", raising=False)
    else:
        monkeypatch.setenv("PBWTAMD_TEAM", "1" if team == "1" else "0")
        monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if team == "onepass" else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")        # grid.y = panel on the one-launch round
    if team == "onepass" and M > 524288:
        pytest.skip("the one-launch round takes up to 1 024 tiles")
    st = torch.cuda.Stream()
    engs = [amd.Engine(M, batch_sites=B, stream=st.cuda_stream) for _ in range(P)]
    bufs = [torch.zeros((N, engs[0].wpc), dtype=torch.int32, device="cuda") for _ in range(P)]
    torch.cuda.synchronize()
    for p in range(P):
        engs[p].synth_device(bufs[p].data_ptr(), 0, N, seed=900 + 17 * p + M, kind=p % 2)
        engs[p].sync()
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    for e in engs:
        e.pass_begin(N)
    step = 8 * ((B + 40) // 8)                              # not a multiple of the batch: full and partial batches in one call
    k = 0
    while k < N:
        n = min(step, N - k)
        amd.pass_advance_many(engs, [b.data_ptr() + k * engs[0].wpc * 4 for b in bufs], n, min(n + 8, N - k), opts)
        k += n
    for p in range(P):
        engs[p].pass_end(opts)
        bits = bufs[p].cpu().numpy().view(np.uint32)
        o = orc.build_bitcols(bits, M, with_d=True)
        a, d = engs[p].get_state()
        assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"]), p
        assert np.array_equal(engs[p].get_packed(), o["yz"]), p
        if M <= 100000:
            assert np.array_equal(engs[p].get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1]), p


def test_match_sweep_sparse_golden_and_no_match_branch(amd, orc):
    """records captured from the reference itself (tests/golden/sparse_sweep.npz), incl. sites where no panel
    haplotype carries the query's allele; and nSparse larger than the batch is refused"""
    g = np.load(os.path.join(GOLDEN, "sparse_sweep.npz"))
    Mp, Mq, N = (int(v) for v in g["nomatch_shape"])
    eng = amd.Engine(Mp, batch_sites=8)
    for nS in (1, 2, 3):
        got, gn, _ = eng.match_sweep_sparse(g["nomatch_pz"], N, g["nomatch_qz"], Mq, nS)
        assert np.array_equal(got, g["nomatch_s%d" % nS].view(got.dtype).reshape(-1)) and gn > 0
    m = np.load(os.path.join(GOLDEN, "mosaic_M300_N400_k0.npz"))
    M, N, Mq = int(m["M"]), int(m["N"]), int(m["Mq"])
    eng = amd.Engine(M - Mq, batch_sites=64)
    got, _, _ = eng.match_sweep_sparse(m["pz"], N, m["qz"], Mq, 3)
    assert np.array_equal(got, g["mosaic_M300_s3"].view(got.dtype).reshape(-1))
    with pytest.raises(amd.PbwtAmdError):
        amd.Engine(M - Mq, batch_sites=4).match_sweep_sparse(m["pz"], N, m["qz"], Mq, 9)


@pytest.mark.parametrize("skel_read", ["1", "0", "onepass", "scan"])
@pytest.mark.parametrize("M,N,batch,kind", [(3000, 200, 64, 0), (1025, 96, 24, 1), (70001, 80, 40, 0), (2, 33, 8, 1), (300000, 24, 8, 0), (600100, 16, 8, 0)])
def test_read_side_both_chains(amd, orc, skel_read, M, N, batch, kind, monkeypatch):
    """the read side (ForwardsReadAD over a packed panel, the reference's -read ... -maxWithin path): the skeleton chain
    with keys derived from the sorted columns through the LF-mapping (default) and the one-site-per-launch chain
    (PBWTAMD_SKEL_READ=0) — a, d, y at every site, histogram, records, -longWithin against the oracle"""
    if skel_read == "onepass" and M > 524288:
        pytest.skip("the one-launch round takes up to 1 024 tiles")
    monkeypatch.setenv("PBWTAMD_SKEL_READ", "0" if skel_read == "0" else "1")
    monkeypatch.setenv("PBWTAMD_ONEPASS", "1" if skel_read in ("onepass", "scan") else "0"); monkeypatch.setenv("PBWTAMD_ONEPASS_MAXW", "1024")    # the one-launch round on the read side: totals from the LF-mapped key rows
    if skel_read == "scan":                                # ... and its scanner form (round 6) at every width
        monkeypatch.setenv("PBWTAMD_ONEPASS_SCAN", "1"); monkeypatch.setenv("PBWTAMD_ONEPASS_SCAN_MIN", "0")
    bits = orc.synth_bitcols(M, N, seed=3000 + M, kind=kind)
    yz = orc.build_bitcols(bits, M, with_d=False)["yz"]
    eng = amd.Engine(M, batch_sites=batch)
    sw = eng.sweep_AD(yz, N)
    s = orc.sweep_AD(yz, M, N)
    for f, n in (("csum_a", N + 1), ("csum_d", N + 1), ("csum_y", N)):      # y at k == N is the reference's stale column
        assert np.array_equal(sw[f][:n], s[f][:n]), "%s differs first at site %d" % (f, int(np.argmax(sw[f][:n] != s[f][:n])))
    assert np.array_equal(eng.max_within(yz, N, mode="hist"), orc.max_within_hist(yz, M, N)[: N + 1])
    if M <= 3000:
        assert np.array_equal(eng.max_within(yz, N, mode="records"), orc.max_within(yz, M, N))
        assert np.array_equal(eng.long_within(yz, N, 20), orc.long_within(yz, M, N, 20))


def test_chunked_build_from_carried_cursor(amd, orc):
    """a panel built in chunks, each starting from the previous chunk's aFend (what -checkpoint does): .pbwt bytes
    concatenate to the one-shot build's, on the skeleton chain (chunks of 64 and 128 sites) and on the fallback (50)"""
    M, N = 5000, 320
    bits = orc.synth_bitcols(M, N, seed=77, kind=0)
    o = orc.build_bitcols(bits, M, with_d=True)
    for chunk in (64, 128, 50):
        eng = amd.Engine(M, batch_sites=64)
        a, yz = None, []
        for k0 in range(0, N, chunk):
            b = eng.build(bits[k0:k0 + chunk], with_d=True, aFstart=a)
            a = b["aFend"]; yz.append(b["yz"])
        assert np.array_equal(np.concatenate(yz), o["yz"]) and np.array_equal(a, o["aFend"])


@pytest.mark.parametrize("path", golden_panels() + [os.path.join(GOLDEN, "merge1.npz")])
def test_cursor_at_is_the_reference_PbwtCursor(amd, orc, path):
    """pbwtamd_cursor_at: every field of the reference's PbwtCursor (pbwt.h:74-87) before site k — a, d, y (stale at
    k == N), c, u, and the byte offsets n / nBlockStart into yz — against the dumps the reference itself produced"""
    g = np.load(path)
    M, N = int(g["M"]), int(g["N"])
    aF = g["aFstart"] if "aFstart" in g else None
    eng = amd.Engine(M, batch_sites=16)
    yz = g["yz"]
    starts = [0]
    for k in range(N):
        starts.append(starts[-1] + orc.unpack3(yz[starts[-1]:], M)[1])
    for k in sorted({0, 1, min(7, N), min(16, N), N // 2, N - 1, N}):
        c = eng.cursor_at(yz, N, k, aFstart=aF)
        assert np.array_equal(c["a"], g["sweep_a"][k]) and np.array_equal(c["d"], g["sweep_d"][k])
        assert np.array_equal(c["y"], g["sweep_y"][k]) and c["c"] == int(g["sweep_c"][k])
        assert np.array_equal(c["u"], np.concatenate([[0], np.cumsum(1 - c["y"].astype(np.int32))]))
        ky = min(k, N - 1)
        assert c["nBlockStart"] == starts[ky] and c["n"] == (starts[k + 1] if k < N else len(yz))
