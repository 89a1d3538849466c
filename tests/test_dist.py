"""CPU: the N>1 path of bench.py (one process per rank, independent panels per rank, barrier +
max-over-ranks timing) with the gloo backend and world_size 2 / 3."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    rank, world = pd.init("gloo")
    units = pd.units_for_rank(int(os.environ["N_UNITS"]), rank, world)
    seeds = [pd.panel_seed(1000, u) for u in units]
    pd.barrier()
    elapsed = 0.01 * (rank + 1)                      # rank-dependent "work"
    worst = pd.max_over_ranks(elapsed)
    total_units = pd.sum_over_ranks(len(units))
    pd.barrier()
    with open(os.path.join(os.environ["OUT_DIR"], "rank" + str(rank) + ".json"), "w") as f:      # per-rank file: stdout lines of ranks can interleave
        json.dump({"rank": rank, "world": world, "units": units, "seeds": seeds, "worst": worst, "total": total_units}, f)
    pd.finish()
""") % ROOT


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,n_units", [(2, 2), (3, 8)])
def test_independent_panels_per_rank_gloo(world, n_units, tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, N_UNITS=str(n_units), OUT_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / ("rank%d.json" % rk))) for rk in range(world)]
    assert sorted(o["rank"] for o in outs) == list(range(world))
    all_units = sorted(u for o in outs for u in o["units"])
    assert all_units == list(range(n_units))                          # every panel exactly once
    assert len({s for o in outs for s in o["seeds"]}) == n_units      # distinct panels
    for o in outs:
        assert o["world"] == world
        assert abs(o["worst"] - 0.01 * world) < 1e-9                  # max over ranks, same on every rank
        assert o["total"] == n_units


def test_units_for_rank_partition():
    from pbwt_amd.dist import units_for_rank
    for world in (1, 2, 4, 8):
        for n in (world, 22, 23):
            got = [u for r in range(world) for u in units_for_rank(n, r, world)]
            assert got == list(range(n))
            sizes = [len(units_for_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


SHARD_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from pbwt_amd import dist as pd
    from pbwt_amd.sharded import sharded_step_AD, owner_ranges
    import oracle
    rank, world = pd.init("gloo")
    M, N, kind = int(os.environ["SH_M"]), int(os.environ["SH_N"]), int(os.environ["SH_KIND"])
    bits = oracle.synth_bitcols(M, N, seed=4242, kind=kind)
    hap = oracle.unpack_bitcols(bits, M)
    want = oracle.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    b = owner_ranges(M, world)
    lo, hi = b[rank], b[rank + 1]
    import torch
    a = torch.arange(lo, hi, dtype=torch.int64)                     # CPU tensors under gloo; the same calls take device tensors under RCCL
    d = torch.zeros(hi - lo, dtype=torch.int64)
    if rank == 0 and hi > lo:
        d[0] = 1
    ok = True
    for k in range(N):
        y = torch.from_numpy(hap[k].astype(np.int64))[a]
        a, d = sharded_step_AD(a, d, y, k, M)
        ok &= bool(np.array_equal(a.numpy(), want["a_dump"][k + 1][lo:hi])) and bool(np.array_equal(d.numpy(), want["d_dump"][k + 1][lo:hi]))
    with open(os.path.join(os.environ["OUT_DIR"], "shard" + str(rank) + ".json"), "w") as f:
        json.dump({"rank": rank, "ok": ok, "n": int(hi - lo)}, f)
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,M,N,kind", [(2, 101, 60, 1), (3, 400, 80, 0), (2, 64, 40, 0)])
def test_position_sharded_step_protocol_gloo(world, M, N, kind, tmp_path):
    """all-gather of carry tuples + all-to-all of (pos, a, d') reproduces the oracle's
    a[] and d[] at every site on every rank's shard"""
    import json
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), SH_M=str(M), SH_N=str(N), SH_KIND=str(kind))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = [json.load(open(tmp_path / ("shard%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs) and sum(o["n"] for o in outs) == M


def test_site_block_plan():
    """pbwt_amd/siteblock.py: the blocks cover [0, N) in order, boundaries sit on batch multiples, and the geometric sizes
    balance the ranks under the cost model (chain-only prefix at rho per site + the block at 1)"""
    from pbwt_amd.siteblock import plan_blocks, model_time
    for N in (1000000, 200000, 4096, 512, 0):
        for world in (1, 2, 3, 4, 8):
            for rho in (0.2, 0.48, 0.84):
                b = plan_blocks(N, world, rho, align=512)
                assert len(b) == world and b[0][0] == 0 and b[-1][1] == N
                assert all(b[g][1] == b[g + 1][0] for g in range(world - 1)) and all(lo <= hi for lo, hi in b)
                assert all(lo % 512 == 0 for lo, _ in b)
                if N >= 100000:
                    t = model_time(b, rho)
                    assert max(t) - min(t) <= 2 * 512 + 1e-9                          # balanced to the alignment
                    assert max(t) <= N * rho / (1 - (1 - rho) ** world) + 2 * 512   # the closed form of the docstring


BLOCK_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    from pbwt_amd import siteblock as sb
    rank, world = pd.init("gloo")
    N = 4096
    blocks = sb.plan_blocks(N, world, 0.5, align=512)
    lo, hi = blocks[rank]
    hist = np.zeros(N + 1, np.int64); hist[lo:hi] = np.arange(lo, hi) + 1                 # each rank contributes its own sites
    if rank == world - 1: hist[N] = 7
    total = sb.reduce_hist(hist)
    yz = sb.gather_packed(np.arange(lo, hi, dtype=np.int64).astype(np.uint8), dst=0)      # site order = rank order
    ok = bool(np.array_equal(total[:N], np.arange(N) + 1)) and total[N] == 7
    if rank == 0:
        ok = ok and bool(np.array_equal(yz, np.arange(N, dtype=np.int64).astype(np.uint8)))
    else:
        ok = ok and yz is None
    with open(os.path.join(os.environ["OUT_DIR"], "blk" + str(rank) + ".json"), "w") as f:
        json.dump({"rank": rank, "ok": bool(ok)}, f)
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_site_block_combine_gloo(world, tmp_path):
    """the two collectives of the site-block mode — all-reduce of the histogram, gather of the packed blocks in rank order"""
    import json
    script = tmp_path / "blk_worker.py"
    script.write_text(BLOCK_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert all(json.load(open(tmp_path / ("blk%d.json" % rk)))["ok"] for rk in range(world))


QS_MERGE_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    from pbwt_amd import queryshard as qs
    rank, world = pd.init("gloo")
    z = np.load(os.environ["QS_IN"])
    N, nS, Mq = int(z["N"]), int(z["nS"]), int(z["Mq"])
    lo, hi = qs.plan_ranges(Mq, world)[rank]
    tagged = z["tagged"]                                   # the full stream, rank-tagged as a device would tag it
    mine = tagged[(tagged[:, 0] >= lo) & (tagged[:, 0] < hi)]            # this rank's share, in its own emission order
    parts = qs._gather_rows(mine)
    streams = []
    for a in parts:
        s = np.zeros(len(a), qs.MERGED_DTYPE)
        for i, f in enumerate(("ai", "bi", "start", "end", "sparse")):
            s[f] = a[:, i]
        streams.append(s)
    merged = qs.merge_streams(streams, N, nS)
    want = z["want"]
    ok = len(merged) == len(want) and all(np.array_equal(merged[f], want[:, i]) for i, f in enumerate(("ai", "bi", "start", "end", "sparse")))
    total = pd.sum_over_ranks(len(mine))
    with open(os.path.join(os.environ["OUT_DIR"], "qs" + str(rank) + ".json"), "w") as f:
        json.dump({"rank": rank, "ok": bool(ok), "total": total, "n": len(want)}, f)
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,nS", [(2, 0), (3, 3)])
def test_query_shard_merge_gloo(world, nS, tmp_path):
    """the host side of -matchDynamic sharded by queries (pbwt_amd/queryshard.py) under gloo: every rank contributes the
    records of its query range tagged with the query's PBWT rank, the all-gather + one stable sort give back exactly
    the oracle's stream — dense and sparse reports, tails at N cursor by cursor"""
    import json
    import numpy as np
    import oracle as orc
    Mp, Mq, N = 60, 17, 50
    rng = np.random.default_rng(5 + world)
    hap = (rng.random((N, Mp + Mq)) < 0.35).astype(np.uint8)
    for k in range(1, N):                                  # some linkage, so that matches have length
        keep = rng.random(Mp + Mq) < 0.7
        hap[k, keep] = hap[k - 1, keep]
    pz = orc.build_bitcols(orc.pack_bitcols(hap[:, :Mp]), Mp, with_d=False)["yz"]
    qb = orc.build_bitcols(orc.pack_bitcols(hap[:, Mp:]), Mq, with_d=False, dump_sites=range(N))
    want, _, _ = orc.match_sweep_sparse(pz, Mp, qb["yz"], Mq, N, nS)
    assert len(want) > 50
    # the query's rank in the query panel's order at the record's site (the final order for the tails at N)
    pos = np.zeros((N + 1, Mq), np.int64)
    for k in range(N):
        pos[k, qb["a_dump"][k]] = np.arange(Mq)
    pos[N, qb["a_final"]] = np.arange(Mq)
    w = np.stack([want[f] for f in ("ai", "bi", "start", "end", "sparse")], axis=1).astype(np.int32)
    tagged = w.copy()
    tagged[:, 4] |= (pos[w[:, 3], w[:, 0]] << 1).astype(np.int32)
    np.savez(tmp_path / "in.npz", want=w, tagged=tagged, N=N, nS=nS, Mq=Mq)
    script = tmp_path / "qs_merge_worker.py"
    script.write_text(QS_MERGE_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), QS_IN=str(tmp_path / "in.npz"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("qs%d.json" % rk))) for rk in range(world)]
    assert all(x["ok"] for x in res) and all(x["total"] == len(want) for x in res)


def test_query_shard_plan():
    from pbwt_amd import queryshard as qs
    for Mq, world in ((10000, 8), (7, 3), (2, 4), (0, 2)):
        r = qs.plan_ranges(Mq, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == Mq
        assert all(r[g][1] == r[g + 1][0] for g in range(world - 1))
        sizes = [hi - lo for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1


ROUND8_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    from pbwt_amd.sharded import sharded_round8, owner_ranges
    import oracle
    import torch
    rank, world = pd.init("gloo")
    M, N, kind = int(os.environ["SH_M"]), int(os.environ["SH_N"]), int(os.environ["SH_KIND"])
    bits = oracle.synth_bitcols(M, N, seed=777, kind=kind)
    hap = oracle.unpack_bitcols(bits, M).astype(np.int64)
    want = oracle.build_bitcols(bits, M, with_d=True, dump_sites=range(0, N + 1, 8))
    b = owner_ranges(M, world)
    lo, hi = b[rank], b[rank + 1]
    a = torch.arange(lo, hi, dtype=torch.int64)
    d = torch.zeros(hi - lo, dtype=torch.int64)
    if rank == 0 and hi > lo:
        d[0] = 1
    ok = True
    for k in range(0, N, 8):
        key_by_hap = sum(hap[k + j] << j for j in range(8))             # every rank holds the panel's columns (replicated, M / 8 bytes per site)
        key = torch.from_numpy(key_by_hap)[a]
        a, d = sharded_round8(a, d, key, k, M)
        r = k // 8 + 1
        ok &= bool(np.array_equal(a.numpy(), want["a_dump"][r][lo:hi])) and bool(np.array_equal(d.numpy(), want["d_dump"][r][lo:hi]))
    with open(os.path.join(os.environ["OUT_DIR"], "r8_" + str(rank) + ".json"), "w") as f:
        json.dump({"rank": rank, "ok": ok, "n": int(hi - lo)}, f)
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,M,N,kind", [(2, 101, 64, 1), (3, 400, 80, 0), (2, 64, 40, 0), (3, 5, 16, 1)])
def test_position_sharded_round_of_eight_sites_gloo(world, M, N, kind, tmp_path):
    """the skeleton round with a rank in the role of a tile (pbwt_amd/sharded.py::sharded_round8): one all-gather of the
    per-key rows + one all-to-all per EIGHT sites reproduce the oracle's a[] and d[] after every round on every shard"""
    import json
    script = tmp_path / "r8_worker.py"
    script.write_text(ROUND8_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), SH_M=str(M), SH_N=str(N), SH_KIND=str(kind))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    outs = [json.load(open(tmp_path / ("r8_%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs) and sum(o["n"] for o in outs) == M


POSSHARD_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from pbwt_amd import dist as pd
    from pbwt_amd import posshard as ps
    import oracle
    rank, world = pd.init("gloo")
    M, N, B = 700, int(os.environ["PS_N"]), 64
    bits = oracle.synth_bitcols(M, N, seed=31, kind=0)
    o = oracle.build_bitcols(bits, M, with_d=True)
    # per-column byte offsets of the oracle's yz: decode run lengths column by column (pack3: every byte is one run, pbwtCore.c:232-238)
    def runlen(b):
        b &= 0x7f
        return b if b < 64 else ((b - 64) << 6 if b < 96 else (b - 96) << 11)
    off = [0]; acc = 0
    for i, byte in enumerate(o["yz"]):
        acc += runlen(int(byte))
        if acc == M:
            off.append(i + 1); acc = 0
    assert len(off) == N + 1

    class StubEngine:                       # the host-visible face of one rank of a sharded pass: what this rank's device consumed
        def __init__(self):
            self.blocks = []; self.connected = None
            k = 0
            while k < N:
                nb = min(B, N - k)
                if nb %% 8 == 0:
                    lo, hi = ps.plan_rounds(nb // 8, world)[rank]
                    if hi > lo: self.blocks.append((k + 8 * lo, 8 * (hi - lo)))
                elif rank == world - 1:     # replicated batch: the last rank's
                    self.blocks.append((k, nb))
                k += nb
        def shard_init(self, r, w): return bytes([r]) * 320
        def shard_connect(self, blobs): self.connected = blobs
        def shard_blocks(self):
            ends, acc = [], 0
            for s0, ns in self.blocks:
                acc += off[s0 + ns] - off[s0]; ends.append(acc)
            return np.array([b[0] for b in self.blocks], np.int64), np.array([b[1] for b in self.blocks], np.int64), np.array(ends, np.int64)
        def get_packed(self):
            return np.concatenate([o["yz"][off[s0]: off[s0 + ns]] for s0, ns in self.blocks] + [np.zeros(0, np.uint8)])
        def get_checksums(self, k0, n):
            mine = np.zeros(N + 1, bool)
            for s0, ns in self.blocks: mine[s0: s0 + ns] = True
            mine[N] = rank == world - 1
            z = lambda v: np.where(mine, v, np.uint64(0))[k0: k0 + n]
            return z(o["csum_a"]), z(o["csum_d"]), z(o["csum_a"])

    eng = StubEngine()
    ps.setup(eng, rank, world)
    ok = eng.connected == [bytes([r]) * 320 for r in range(world)]
    yz = ps.gather_packed(eng)
    cs = ps.gather_checksums(eng, 0, N + 1)
    hist = ps.reduce_hist(np.full(5, rank + 1, np.int64))
    ok = ok and bool(np.array_equal(hist, np.full(5, world * (world + 1) // 2)))
    if rank == 0:
        ok = ok and bool(np.array_equal(yz, o["yz"])) and bool(np.array_equal(cs[0], o["csum_a"])) and bool(np.array_equal(cs[1], o["csum_d"]))
    with open(os.path.join(os.environ["OUT_DIR"], "psd" + str(rank) + ".json"), "w") as f:
        json.dump({"ok": bool(ok), "blocks": len(eng.blocks)}, f)
    pd.finish()
""") % ROOT


@pytest.mark.parametrize("world,N", [(2, 200), (3, 333)])
def test_position_shard_host_side_gloo(world, N, tmp_path):
    """pbwt_amd/posshard.py under gloo at world_size 2 / 3: the handle all-gather, the interleaving of the ranks' pack3 blocks
    (sharded batches by rounds, ragged batches on the last rank) into the oracle's yz, checksums and histograms adding up"""
    import json
    script = tmp_path / "psd_worker.py"
    script.write_text(POSSHARD_WORKER)
    env = dict(os.environ, OUT_DIR=str(tmp_path), PS_N=str(N))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    outs = [json.load(open(tmp_path / ("psd%d.json" % rk))) for rk in range(world)]
    assert all(o["ok"] for o in outs), outs


def test_position_shard_plans():
    from pbwt_amd import posshard as ps
    for world in (1, 2, 3, 8):
        for n in (world, 7, 64, 1954):
            tb = ps.tile_bounds(n, world)
            assert tb[0] == 0 and tb[-1] == n and all(b >= a for a, b in zip(tb, tb[1:]))
            if n >= world:
                assert all(b > a for a, b in zip(tb, tb[1:]))          # every rank owns at least one tile
            pr = ps.plan_rounds(n, world)
            assert pr[0][0] == 0 and pr[-1][1] == n and all(a[1] == b[0] for a, b in zip(pr, pr[1:]))
    import numpy as np
    with pytest.raises(ValueError, match="tile the sites"):
        ps.merge_packed([(np.array([0]), np.array([8]), np.array([3]), np.zeros(3, np.uint8)), (np.array([16]), np.array([8]), np.array([2]), np.zeros(2, np.uint8))])
