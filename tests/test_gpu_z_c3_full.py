"""GPU (-m gpu): BASELINE.json configs[2] ITSELF — 100 000 haplotypes x 1 000 000 sites, build + -maxWithin — run end to end
under the oracle, on the exact pass bench.py times (same seed, same option set OPT_WITH_D | OPT_WITHIN_HIST | OPT_PACK3, same
pbwtamd_pass_advance calls of 50 000 sites with 8 look-ahead columns).

The oracle takes ~8 minutes of one core for this panel, so it is checked BLOCK BY BLOCK from the device's own checkpoints: at
every cut the uninterrupted pass hands out its cursor (a_k, d_k: pbwtamd_get_state), its cumulative histogram and the pack3
bytes written since the previous cut; oracle.segment() (orc_segment: the reference's build loop pbwtIO.c:477-483 +
matchMaximalWithin pbwtMatch.c:115-142 + WriteForwardsAD pbwtCore.c:580-585, continued from a given cursor) runs each block
from checkpoint i on its own host thread and must arrive at checkpoint i+1 with the same bytes and the same histogram
increment.  With every block checked ("full": the default where the host has >= 8 CPUs; 20 blocks, ~30 s each, in parallel)
that is an induction over the whole panel: every site's column bytes, the final state and the full histogram of configs[2]
are the oracle's.  On a small host ("windows") only 512-site windows are checked — at every 50 000 sites plus windows across
sites 32 767 and 65 535, where the 16-bit hand-off slots escape to the 32-bit ring (L >= 32 767) — as VERDICT r5 item 1b asks.
PBWT_C3_VERIFY=full|windows overrides.  The histogram total is pinned in tests/golden/c3_full.json, which bench.py compares
its own line with (`verified_hist_total`)."""
import json
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M, N, STEP, SEED, KIND = 100000, 1000000, 50000, 0x5EED0001, 0


def plan_cuts(mode):
    """sites at which the pass hands out a checkpoint, and which blocks [cut i, cut i+1) the oracle re-runs"""
    if mode == "full":
        cuts = list(range(0, N, STEP)) + [N]
        return cuts, [True] * (len(cuts) - 1)
    starts = sorted(set(list(range(0, N - 512, STEP)) + [32767 - 256, 65535 - 256, N - 512]))
    cuts, verify = [0], []
    for s0 in starts:
        if s0 > cuts[-1]:
            cuts.append(s0); verify.append(False)
        cuts.append(s0 + 512); verify.append(True)
    if cuts[-1] != N:
        cuts.append(N); verify.append(False)
    return cuts, verify


def test_config2_itself_full_length_under_the_oracle(gpu_lib, orc):
    import torch
    amd = gpu_lib
    ncpu = os.cpu_count() or 1
    mode = os.environ.get("PBWT_C3_VERIFY", "full" if ncpu >= 8 else "windows")
    cuts, verify = plan_cuts(mode)
    eng = amd.Engine(M, batch_sites=512)
    buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng.synth_device(buf.data_ptr(), 0, N, seed=SEED, kind=KIND); eng.sync()
    opts = amd.OPT_WITH_D | amd.OPT_WITHIN_HIST | amd.OPT_PACK3
    ptr = lambda k: buf.data_ptr() + k * eng.wpc * 4

    # ---- the device: ONE pass, site 0 to site N; advance calls end at the bench's step boundaries and at the cuts
    a0 = np.arange(M, dtype=np.int32); d0 = np.zeros(M + 1, np.int32); d0[0] = d0[M] = 1
    cps = [dict(k=0, a=a0, d=d0, hist=np.zeros(N + 2, np.int64), yz=None)]
    eng.pass_begin(N)
    t0 = time.perf_counter()
    k = 0
    for cut in cuts[1:]:
        while k < cut:
            n = min(cut, (k // STEP + 1) * STEP) - k
            eng.pass_advance(ptr(k), n, min(n + 8, N - k), opts)
            k += n
        if cut == N:
            eng.pass_end(opts)                              # the k == N sweep
        a, d = eng.get_state()
        hist = eng.get_hist(N + 2)                          # flushes the pending consumers: cumulative over the sites < cut
        cps.append(dict(k=cut, a=a, d=d, hist=hist, yz=eng.drain_packed().copy()))
    t_dev = time.perf_counter() - t0
    assert eng.chain_timing()[1] > 0

    # ---- the oracle: every verified block from checkpoint i must arrive at checkpoint i+1
    lock = threading.Lock()

    def check(i):
        c0, c1 = cps[i], cps[i + 1]
        with lock:                                          # one device-to-host copy at a time
            bits = buf[c0["k"]:c1["k"]].cpu().numpy().view(np.uint32)
        s = orc.segment(bits, M, c0["k"], N, c0["a"], c0["d"], yz_cap=c1["yz"].size + M + 16)
        dh = c1["hist"] - c0["hist"]
        bad = []
        if not np.array_equal(s["a"], c1["a"]): bad.append("a[] at site %d" % c1["k"])
        if not np.array_equal(s["d"], c1["d"]): bad.append("d[] at site %d" % c1["k"])
        if not np.array_equal(s["yz"], c1["yz"]): bad.append("pack3 bytes (%d vs %d)" % (s["yz"].size, c1["yz"].size))
        if not np.array_equal(s["hist"], dh): bad.append("maxWithin histogram at length %d" % int(np.argmax(s["hist"] != dh)))
        return (c0["k"], c1["k"], bad)

    todo = [i for i, v in enumerate(verify) if v]
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(len(todo), ncpu, 32))) as ex:
        res = list(ex.map(check, todo))
    t_orc = time.perf_counter() - t1
    failed = [(k0, k1, bad) for (k0, k1, bad) in res if bad]
    assert not failed, "blocks that differ from the oracle: %r" % failed[:4]

    # ---- whole-pass facts
    a, d = cps[-1]["a"], cps[-1]["d"]
    assert np.array_equal(np.sort(a), np.arange(M)) and d[0] == N + 1 and d[M] == N + 1
    total = int(cps[-1]["hist"].sum())
    nbytes = int(sum(c["yz"].size for c in cps[1:]))
    sites_checked = int(sum(cps[i + 1]["k"] - cps[i]["k"] for i in todo))
    report = {"haplotypes": M, "sites": N, "kind": KIND, "seed": SEED, "step": STEP, "within_reports_hist_total": total, "pack3_bytes": nbytes,
              "mode": mode, "blocks_checked": len(todo), "sites_checked_against_oracle": sites_checked,
              "device_seconds_incl_checkpoints": t_dev, "oracle_seconds_wall": t_orc, "oracle_threads": max(1, min(len(todo), ncpu, 32))}
    out_dir = os.path.join(__import__("conftest").ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        json.dump(report, open(os.path.join(out_dir, "c3_full.json"), "w"), indent=1)
    except OSError:
        pass
    gpath = os.path.join(__import__("conftest").GOLDEN, "c3_full.json")
    if os.path.exists(gpath):                               # pinned by a full run; bench.py compares its line with it
        g = json.load(open(gpath))
        assert (g["haplotypes"], g["sites"], g["seed"]) == (M, N, SEED)
        assert total == g["within_reports_hist_total"] and nbytes == g["pack3_bytes"]
    if mode == "full":
        assert sites_checked == N
