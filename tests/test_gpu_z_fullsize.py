"""GPU (-m gpu): BASELINE.json configurations at full size.

configs[1] (10k haplotypes x 100k sites, build with ForwardsAD): every site's a[] and d[] against the
oracle through order-sensitive checksums, plus the final arrays and the packed bytes.
configs[2] scale (100k haplotypes): size-independent properties that tie independent code paths
together — the build-side chain (two sites per launch, gather mode) and the read-side chain (one site
per launch, sorted mode) must produce the same a/d at every site, decode(encode(panel)) == panel,
a[] stays a permutation, the divergence sentinels hold.  (configs[2] and configs[4] against the oracle at their own
width: tests/test_gpu_z_configs.py.)"""
import os

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


def test_config1_full_size_build_AD(gpu_lib, orc):
    amd = gpu_lib
    M, N = 10000, 100000
    eng = amd.Engine(M, batch_sites=512)
    buf = device_panel(eng, N, seed=0xC0FFEE)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_PACK3
    eng.pass_begin(N)
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    a, d = eng.get_state()
    ca, cd, _ = eng.get_checksums(0, N + 1)
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True, want_yz=False)
    assert np.array_equal(ca, o["csum_a"]), "a[] differs at site %d" % int(np.argmax(ca != o["csum_a"]))
    assert np.array_equal(cd, o["csum_d"]), "d[] differs at site %d" % int(np.argmax(cd != o["csum_d"]))
    assert np.array_equal(a, o["aFend"]) and np.array_equal(d, o["d_final"])
    # two sites per launch all the way; on the skeleton path ONE launch per 8 sites (PBWTAMD_ONEPASS=0: TWO at this width (40 tiles of 256 positions:
    # the rank kernel scans the tile table itself up to 48 tiles), three above
    per_round = 2 if (M + 255) // 256 <= int(os.environ.get("PBWTAMD_SKN_MAXW", "48")) and os.environ.get("PBWTAMD_SKN", "1") != "0" else 3
    if os.environ.get("PBWTAMD_ONEPASS", "1") != "0":
        per_round = 1                          # (round 5) the one-launch round: totals precomputed per batch, the tile prefixes fetched inside the launch
    assert eng.chain_timing()[1] == (per_round * (N // 8) if os.environ.get("PBWTAMD_SKEL", "1") != "0" else N // 2)


def test_config2_scale_cross_path_properties(gpu_lib, orc):
    amd = gpu_lib
    M, N = 100000, 3000
    eng = amd.Engine(M, batch_sites=512)
    buf = device_panel(eng, N, seed=0x5EED0001)
    bits = buf.cpu().numpy().view(np.uint32)
    b = eng.build(bits, with_d=True)                       # build side: gather mode, two sites per launch
    # a[] is a permutation; sentinels d[0] = d[M] = N + 1; interior divergences are valid start positions
    assert np.array_equal(np.sort(b["aFend"]), np.arange(M))
    assert b["dFend"][0] == N + 1 and b["dFend"][M] == N + 1 and b["dFend"][1:M].max() <= N and b["dFend"].min() >= 0
    # decode(encode(panel)) == panel, through the read-side chain (sorted mode, one site per launch)
    hap = eng.haplotypes(b["yz"], N)
    assert np.array_equal(hap, orc.unpack_bitcols(bits, M))
    # both chains give the same a/d at every site (checksums), and the same final state
    eng.pass_begin(N)
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM
    eng.pass_advance(buf.data_ptr(), N, N, opts)
    eng.pass_end(opts)
    ca, cd, cy = eng.get_checksums(0, N + 1)
    sw = eng.sweep_AD(b["yz"], N)
    assert np.array_equal(sw["csum_a"], ca) and np.array_equal(sw["csum_d"], cd) and np.array_equal(sw["csum_y"][:N], cy[:N])
    # pack3 codec round trip on the device at this width
    sorted_cols = eng.unpack3(b["yz"], N)
    assert np.array_equal(eng.pack3(sorted_cols), b["yz"])
    # the oracle on a prefix of the same panel: every site's a/d and the maxWithin histogram of the prefix panel
    n0 = 1024
    o = orc.build_bitcols(bits[:n0], M, with_d=True)
    assert np.array_equal(ca[: n0 + 1], o["csum_a"]) and np.array_equal(cd[: n0 + 1], o["csum_d"])
    assert np.array_equal(eng.max_within(o["yz"], n0, mode="hist"), orc.max_within_hist(o["yz"], M, n0)[: n0 + 1])


def test_million_haplotypes_short_panel(gpu_lib, orc):
    """north-star width (M = 1M): two-site launches with four positions per thread, a few sites, full
    arrays against the oracle"""
    amd = gpu_lib
    M, N = 1000000, 24
    eng = amd.Engine(M, batch_sites=8)
    buf = device_panel(eng, N, seed=11)
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    b = eng.build(bits, with_d=True)
    assert np.array_equal(b["aFend"], o["aFend"]) and np.array_equal(b["dFend"], o["d_final"]) and np.array_equal(b["yz"], o["yz"])
    sw = eng.sweep_AD(o["yz"], N)
    s = orc.sweep_AD(o["yz"], M, N)
    assert np.array_equal(sw["csum_a"], s["csum_a"]) and np.array_equal(sw["csum_d"], s["csum_d"])
