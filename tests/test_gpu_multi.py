"""GPU (-m gpu): the multi-GPU device paths at world_size 2 on the ONE GPU of the test box — two processes launched by
torch.distributed.run, each with its own engine on device 0, gloo for the (small) collectives — and the checkpoint
restart they rest on.  On an 8-GPU node the same code runs one rank per GPU over RCCL (bench.py --mode siteblock)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch
    from pbwt_amd import dist as pd
    from pbwt_amd import siteblock as sb
    import pbwt_amd as amd
    rank, world = pd.init("gloo")
    M, N, B = int(os.environ["SB_M"]), int(os.environ["SB_N"]), 512
    eng = amd.Engine(M, batch_sites=B, device=0)
    panel = torch.empty((N, eng.wpc), dtype=torch.int32, device="cuda:0")
    eng.synth_device(panel.data_ptr(), 0, N, seed=0xB10C, kind=int(os.environ["SB_KIND"]))     # every rank holds the panel's columns
    eng.sync()
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    blocks = sb.plan_blocks(N, world, float(os.environ["SB_RHO"]), align=B)
    sb.run_block(eng, lambda k: panel.data_ptr() + k * eng.wpc * 4, N, blocks[rank], opts, is_last=(rank == world - 1), step=2048)
    hist = sb.reduce_hist(eng.get_hist(N + 1))
    yz = sb.gather_packed(eng.get_packed(), dst=0)
    ok = True
    if rank == 0:
        import oracle
        bits = panel.cpu().numpy().view(np.uint32)
        o = oracle.build_bitcols(bits, M, with_d=True, want_csum=False)
        ok = bool(np.array_equal(yz, o["yz"])) and bool(np.array_equal(hist, oracle.max_within_hist(o["yz"], M, N)[: N + 1]))
    with open(os.path.join(os.environ["OUT_DIR"], "sb" + str(rank) + ".json"), "w") as f:
        json.dump({"rank": rank, "ok": bool(ok), "block": list(blocks[rank])}, f)
    eng.close()
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,M,N,kind,rho", [(2, 30000, 4096, 0, 0.6), (3, 5000, 6144, 1, 0.3), (2, 100000, 2048, 0, 0.84)])
def test_site_block_sharding_two_ranks_one_gpu(world, M, N, kind, rho, tmp_path):
    """one panel, G ranks: chain-only prefix + own block with consumers on each rank; summed histogram and concatenated
    pack3 bytes equal the oracle's for the whole panel"""
    script = tmp_path / "sb_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), SB_M=str(M), SB_N=str(N), SB_KIND=str(kind), SB_RHO=str(rho))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    outs = [json.load(open(tmp_path / ("sb%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs), outs
    assert outs[0]["block"][0] == 0 and outs[-1]["block"][1] == N


def test_restart_from_checkpoint(gpu_lib, orc):
    """pbwtamd_pass_begin(a_k, k0) + pbwtamd_pass_set_d(d_k): a pass restarted from the (a, d) another one stopped with
    reproduces the uninterrupted pass — every later site's a/d, the histogram of the remaining sites, the final state"""
    import torch
    amd = gpu_lib
    M, N, k0 = 20000, 1536, 520
    eng = amd.Engine(M, batch_sites=256)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()               # the engine enqueues on its own stream: the fill must have landed
    eng.synth_device(buf.data_ptr(), 0, N, seed=5, kind=0); eng.sync()
    bits = buf.cpu().numpy().view(np.uint32)
    o = orc.build_bitcols(bits, M, with_d=True)
    ptr = lambda k: buf.data_ptr() + k * eng.wpc * 4
    eng.pass_begin(N)
    eng.pass_advance(ptr(0), k0, k0 + 8, amd.OPT_WITH_D)
    a, d = eng.get_state()
    eng.pass_stop()
    opts = amd.OPT_WITH_D | amd.OPT_CHECKSUM | amd.OPT_WITHIN_HIST
    e2 = amd.Engine(M, batch_sites=256)                       # "another rank"
    e2.pass_begin(N, k0=k0, aInit=a)
    e2.pass_set_d(d)
    e2.pass_advance(ptr(k0), N - k0, N - k0, opts)
    e2.pass_end(opts)
    ca, cd, _ = e2.get_checksums(k0, N - k0 + 1)
    assert np.array_equal(ca, o["csum_a"][k0:]) and np.array_equal(cd, o["csum_d"][k0:])
    a2, d2 = e2.get_state()
    assert np.array_equal(a2, o["aFend"]) and np.array_equal(d2, o["d_final"])
    # histogram of the sites k0..N only = the whole panel's minus the first k0 sites' (from an uninterrupted pass stopped at k0)
    eng.pass_begin(N)
    eng.pass_advance(ptr(0), 512, 520, amd.OPT_WITH_D | amd.OPT_WITHIN_HIST)
    eng.pass_advance(ptr(512), 8, 16, amd.OPT_WITH_D | amd.OPT_WITHIN_HIST)
    eng.pass_stop()
    head = eng.get_hist(N + 1)
    assert np.array_equal(head + e2.get_hist(N + 1), orc.max_within_hist(o["yz"], M, N)[: N + 1])
    with pytest.raises(amd.PbwtAmdError, match="sentinels"):
        e2.pass_begin(N, k0=k0, aInit=a); e2.pass_set_d(np.zeros(M + 1, np.int32))


SHARD_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch
    from pbwt_amd import dist as pd
    from pbwt_amd.sharded import sharded_step_AD, owner_ranges
    import oracle
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))     # RCCL; one rank: the collectives run, on device tensors
    rank, world = dist.get_rank(), dist.get_world_size()
    M, N = 5000, 48
    bits = oracle.synth_bitcols(M, N, seed=99, kind=0)
    hap = torch.from_numpy(oracle.unpack_bitcols(bits, M).astype(np.int64)).cuda()
    want = oracle.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    b = owner_ranges(M, world)
    lo, hi = b[rank], b[rank + 1]
    a = torch.arange(lo, hi, dtype=torch.int64, device="cuda")
    d = torch.zeros(hi - lo, dtype=torch.int64, device="cuda")
    if rank == 0: d[0] = 1
    ok = True
    for k in range(N):
        a, d = sharded_step_AD(a, d, hap[k][a], k, M)
        ok &= a.is_cuda and bool(np.array_equal(a.cpu().numpy(), want["a_dump"][k + 1][lo:hi])) and bool(np.array_equal(d.cpu().numpy(), want["d_dump"][k + 1][lo:hi]))
    # the skeleton round with a rank in the role of a tile: two collectives per EIGHT sites
    from pbwt_amd.sharded import sharded_round8
    a = torch.arange(lo, hi, dtype=torch.int64, device="cuda")
    d = torch.zeros(hi - lo, dtype=torch.int64, device="cuda")
    if rank == 0: d[0] = 1
    ok8 = True
    for k in range(0, N, 8):
        key = sum(hap[k + j] << j for j in range(8))[a]
        a, d = sharded_round8(a, d, key, k, M)
        ok8 &= a.is_cuda and bool(np.array_equal(a.cpu().numpy(), want["a_dump"][k + 8][lo:hi])) and bool(np.array_equal(d.cpu().numpy(), want["d_dump"][k + 8][lo:hi]))
    json.dump({"ok": bool(ok), "ok8": bool(ok8)}, open(os.path.join(os.environ["OUT_DIR"], "ps.json"), "w"))
    pd.finish()
""") % ROOT


def test_position_sharded_step_on_device_tensors_over_rccl(tmp_path):
    """pbwt_amd/sharded.py with DEVICE tensors and the RCCL backend ("nccl") on the GPU box: the same function the gloo tests
    drive at world_size 2 and 3 on the CPU.  One rank here (the box has one GPU; RCCL refuses two ranks on one device), so the
    all-gather and the all-to-all run degenerate but through RCCL, and every site's a/d equals the oracle's."""
    script = tmp_path / "ps_worker.py"
    script.write_text(SHARD_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(tmp_path / "ps.json"))
    assert res["ok"] and res["ok8"]


@pytest.mark.parametrize("world,M,N,B,kind,step,csum,k2s_local", [
    (3, 70000, 776, 512, 0, 8192, 1, "0"),    # PBWTAMD_K2S_LOCAL=0: the two-pass tile scan with waiting workgroups (the form before round 5)
    (2, 100000, 1024, 512, 0, 8192, 0, "0"),
    (2, 30000, 1024, 256, 0, 8192, 1, "1"),        # 256-position tiles (M <= 56 k), four batches: both rings reused
    (3, 5000, 520, 128, 1, 200, 1, "1"),           # iid panel, ragged advance calls: replicated (non-multiple-of-8) batches between sharded ones
    (2, 100000, 1024, 512, 0, 8192, 0, "1"),       # configs[2]'s width on the bench option set (packed fill: no checksums)
    (3, 70000, 776, 512, 0, 8192, 1, "1"),         # N not a multiple of 8: the tail runs replicated
    (2, 1000000, 1024, 512, 0, 8192, 0, "1"),      # BASELINE configs[3]'s width (1 M haplotypes), bench option set, two ranks, two batches
    (3, 1000000, 264, 128, 0, 8192, 1, "1"),       # the same width on three ranks with every site's checksum
    (4, 70000, 264, 128, 0, 8192, 1, "1"),         # world 4: 137 tiles of 512 positions -> 34-35 tiles per rank, four ranks' rows folded in front of the last
    (8, 70000, 136, 64, 0, 8192, 1, "1"),          # world 8 = SHARD_MAX: all eight entries of pb[] / tb[], the 7-compare owner search, 8 exchange rows, one round per rank and batch
    (8, 300000, 72, 64, 1, 8192, 0, "1"),          # world 8, iid panel on the bench option set (packed fill), 586 tiles: 73-74 per rank
])
def test_position_sharded_chain_ranks_one_gpu(world, M, N, B, kind, step, csum, k2s_local, tmp_path):
    """BASELINE configs[3]'s device path at small widths: `world` processes on one GPU, each owning a range of positions of the
    sorted order (hipIpc peer stores for the scatter, flag barriers per round), consumers sharded by site inside every batch;
    every site's a/d checksum, the summed histogram, the interleaved pack3 bytes and the final state equal the oracle's"""
    env = dict(os.environ, OUT_DIR=str(tmp_path), PS_M=str(M), PS_N=str(N), PS_B=str(B), PS_KIND=str(kind), PS_STEP=str(step), PS_CSUM=str(csum),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PBWTAMD_K2S_LOCAL=k2s_local)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "posshard_worker.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    outs = [json.load(open(tmp_path / ("ps%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs), outs
    assert outs[0]["range"][0] == 0 and outs[-1]["range"][1] == M


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="needs a node with at least two MI355X: every box this build has seen so far has ONE GPU, so the position-sharded "
                                                "path is single-device-validated only (DESIGN.md section 6.3) — this is the first thing to run on a multi-GPU node")
@pytest.mark.parametrize("M,N,B,kind,csum", [(70000, 264, 128, 0, 1), (300000, 136, 64, 1, 0), (1000000, 1024, 512, 0, 0)])
def test_position_sharded_across_devices(M, N, B, kind, csum, tmp_path):
    """the same check as test_position_sharded_chain_ranks_one_gpu with rank r on DEVICE r (world = min(device_count, 8)): peer stores into hipIpc
    mappings of another GPU's rings, the flag barriers and the per-round row exchange over xGMI.  What one shared GPU cannot show — a peer store that
    is late or stale in the owner's L2 — fails here as a checksum mismatch at the first site it touches."""
    world = min(_device_count(), 8)
    # first the platform probe across the same devices (tools/ipcprobe.hip, memory kind 0 = the plain hipMalloc rings the sharded chain exports): peer scatter,
    # flag barrier, read-back — its "mismatches" are words a peer wrote that the owner did not see.  Printed (pytest -s / the failure text below), so that a
    # visibility failure of the platform is told apart from a defect of the chain at the first multi-GPU run.
    probe = os.path.join(ROOT, "tools", "ipcprobe")
    probe_report = "ipcprobe not built"
    if os.path.exists(probe):
        pr = subprocess.run([probe, str(world), "200", "0"], capture_output=True, text=True, timeout=300, env=dict(os.environ, IPCPROBE_SPREAD="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        lines = [ln for ln in pr.stdout.splitlines() if "mismatches" in ln or "staggered barrier" in ln or ln.startswith("ipcprobe:")]
        stale = sum(int(ln.split("mismatches")[1].split(",")[0]) for ln in lines if "mismatches" in ln)
        probe_report = "ipcprobe across %d devices: rc %d, %d stale words\n%s" % (world, pr.returncode, stale, "\n".join(lines))
    print(probe_report)
    env = dict(os.environ, OUT_DIR=str(tmp_path), PS_M=str(M), PS_N=str(N), PS_B=str(B), PS_KIND=str(kind), PS_STEP="8192", PS_CSUM=str(csum), PS_SPREAD="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "posshard_worker.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, probe_report + "\n" + r.stdout[-2000:] + r.stderr[-4000:]
    outs = [json.load(open(tmp_path / ("ps%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs), (probe_report, outs)
    assert sorted(o["device"] for o in outs) == list(range(world))


QS_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    from pbwt_amd import queryshard as qs
    import pbwt_amd as amd
    rank, world = pd.init("gloo")
    z = np.load(os.environ["QS_IN"])
    Mp, Mq, N, nS = int(z["Mp"]), int(z["Mq"]), int(z["N"]), int(z["nS"])
    eng = amd.Engine(Mp, batch_sites=int(z["batch"]), device=0)
    recs, ev, nom, tot = qs.match_sweep_sharded(eng, z["pz"], N, z["qz"], Mq, nS)
    if rank == 0:
        np.savez(os.path.join(os.environ["OUT_DIR"], "merged.npz"), recs=recs, ev=ev, nom=nom, tot=np.array(tot, dtype=np.int64))
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,Mp,Mq,N,kind,nS,batch", [(2, 3000, 50, 200, 0, 0, 64), (3, 2500, 31, 130, 1, 4, 128), (2, 20000, 300, 260, 0, 0, 512)])
def test_query_sharding_ranks_one_gpu(world, Mp, Mq, N, kind, nS, batch, tmp_path, orc):
    """-matchDynamic with the QUERIES sharded over `world` ranks (pbwt_amd/queryshard.py): every rank sweeps its range of
    queries on the device, the rank-tagged streams merge into exactly the oracle's stream; counts and totals add up"""
    bits = orc.synth_bitcols(Mp + Mq, N, seed=Mp * 7 + N, kind=kind)
    hap = orc.unpack_bitcols(bits, Mp + Mq)
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qz = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False)["yz"]
    want, nomatch, tot = orc.match_sweep_sparse(pz, Mp, qz, Mq, N, nS)
    np.savez(tmp_path / "in.npz", pz=pz, qz=qz, Mp=Mp, Mq=Mq, N=N, nS=nS, batch=batch)
    script = tmp_path / "qs_worker.py"
    script.write_text(QS_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), QS_IN=str(tmp_path / "in.npz"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    m = np.load(tmp_path / "merged.npz")
    got = m["recs"]
    assert len(got) == len(want)
    for f in ("ai", "bi", "start", "end", "sparse"):
        assert np.array_equal(got[f], want[f]), f
    assert int(m["nom"]) == nomatch and tuple(int(v) for v in m["tot"]) == tuple(tot)


def test_query_range_events_and_reset(gpu_lib, orc):
    """pbwtamd_set_query_range: the no-match events of the ranges merge into the full run's log order; resetting the range
    gives the plain stream back (the reference's own golden with sites where no panel haplotype carries the query's allele)"""
    import pbwt_amd as amd
    from pbwt_amd import queryshard as qs
    g = np.load(os.path.join(ROOT, "tests", "golden", "sparse_sweep.npz"))
    Mp, Mq, N = (int(v) for v in g["nomatch_shape"])
    eng = amd.Engine(Mp, batch_sites=8)
    for nS in (1, 3):
        full, fn, ft = eng.match_sweep_sparse(g["nomatch_pz"], N, g["nomatch_qz"], Mq, nS)
        fev = eng.nomatch_events()
        assert fn > 0 and np.array_equal(full, g["nomatch_s%d" % nS].view(full.dtype).reshape(-1))
        parts = [qs.run_range(eng, g["nomatch_pz"], N, g["nomatch_qz"], Mq, lo, hi, nS) for lo, hi in qs.plan_ranges(Mq, 3)]
        merged = qs.merge_streams([p[0] for p in parts], N, nS)
        for f in ("ai", "bi", "start", "end", "sparse"):
            assert np.array_equal(merged[f], full[f]), (nS, f)
        assert sum(p[2] for p in parts) == fn and tuple(sum(p[3][i] for p in parts) for i in (0, 1)) == tuple(ft)
        assert np.array_equal(qs.merge_events([p[1] for p in parts]), fev)
        again, _, _ = eng.match_sweep_sparse(g["nomatch_pz"], N, g["nomatch_qz"], Mq, nS)     # range reset by run_range
        assert np.array_equal(again, full)
